#!/usr/bin/env python3
"""isa_loops.py <file.s> <kernel-name-substring> [first-block last-block]: the basic blocks of one kernel of a hipcc -S
listing with their instruction mix (V vector ALU, S scalar ALU, L LDS, G global / flat memory, B branches, W waits / nops),
their branches, and the loops (backward branches).  With a block range: the listing of those blocks.
How the per-score-step numbers of DESIGN.md section 4 were read off the assembly."""
import re
import sys
from collections import Counter


def cat(ins):
    op = ins.split()[0]
    if op.startswith("v_"):
        return "V"
    if op.startswith("ds_"):
        return "L"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "B"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "W"
    if op.startswith("s_"):
        return "S"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "G"
    return "?"


def kernel_blocks(path, name):
    lines = open(path).read().split("\n")
    start = None
    for i, ln in enumerate(lines):
        if re.match(r"^_Z\w*:", ln) and name in ln.split(":")[0]:
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    blocks = [["entry", []]]
    for ln in lines[start + 1:]:
        if ln.startswith(".Lfunc_end") or re.match(r"^\s*s_endpgm", ln) and False:
            break
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            blocks.append([m.group(1), []])
            continue
        t = ln.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        blocks[-1][1].append(t.split(";")[0].strip())
    return blocks


def main():
    blocks = kernel_blocks(sys.argv[1], sys.argv[2])
    idx = {b[0]: i for i, b in enumerate(blocks)}
    if len(sys.argv) >= 5:
        a, b = int(sys.argv[3]), int(sys.argv[4])
        for i in range(a, b + 1):
            print("%d %s:" % (i, blocks[i][0]))
            for ins in blocks[i][1]:
                print("    " + ins)
        return
    tot = Counter()
    for i, b in enumerate(blocks):
        c = Counter(cat(x) for x in b[1])
        tot.update(c)
        br = []
        for x in b[1]:
            m = re.match(r"s_c?branch(\w*)\s+(\.LBB\d+_\d+)", x)
            if m:
                t = idx.get(m.group(2), -1)
                br.append("%s->%d%s" % (m.group(1).lstrip("_") or "always", t, " (back)" if 0 <= t <= i else ""))
        print("%4d %-12s %4d  %-44s %s" % (i, b[0], len(b[1]), " ".join("%s%d" % (k, c[k]) for k in "VSLGBW?" if c[k]), " ".join(br)))
    print("total", sum(tot.values()), dict(tot))


main()
