#!/usr/bin/env python3
"""Extracts the INPUT DATA of two of the reference's own unit tests into tests/golden/ref_test_inputs.json (run in the build
container only: it reads /root/reference, which does not exist on the GPU box):

  cmd/lib-chaining_test.go:33-89       the 35 seed anchors (QBegin, TBegin, Len) fed to Chainer.Chain
  cmd/lib-seq_compare_test.go:51-52    the two sequences fed to SeqComparator.Index / Compare

The reference's tests only LOG what comes out (no expected values), so SURVEY.md 8c(iv) asks to run these inputs through the
restatement and commit the outputs as goldens: the script does that with the oracle (ClearSubstrPairs + Chain as Search
calls them; Index + Compare) and stores the outputs beside the inputs.  tests/test_reference_test_inputs.py then checks
that the oracle still gives them and that the product's device algorithms (compiled for the host) give the same."""
import ctypes as C
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference/lexicmap/cmd"


def main():
    import oracle as O
    L = O.lib()
    src = open(os.path.join(REF, "lib-chaining_test.go")).read()
    body = src[src.index("subs := []*SubstrPair{"):src.index("tmp := []*SearchResult{")]
    subs = [[int(a), int(b), int(c)] for a, b, c in
            re.findall(r"^\s*\{QBegin:\s*(\d+),\s*TBegin:\s*(\d+),\s*Len:\s*(\d+)\},", body, flags=re.M)]
    assert len(subs) == 35, len(subs)
    src2 = open(os.path.join(REF, "lib-seq_compare_test.go")).read()
    s1 = re.search(r'^\s*s1 := \[\]byte\("([ACGT]+)"\)', src2, flags=re.M).group(1)
    s2 = re.search(r'^\s*s2 := \[\]byte\("([ACGT]+)"\)', src2, flags=re.M).group(1)
    # ---- outputs of the oracle
    K = 31
    oa = (O.Sub * len(subs))()
    for i, (q, t, ln) in enumerate(subs):
        oa[i].qbegin, oa[i].tbegin, oa[i].len = q, t, ln
    n = L.lmo_clear_subs(oa, len(subs), K)
    cleared = [[oa[i].qbegin, oa[i].tbegin, oa[i].len] for i in range(n)]
    coff, cidx, nch = C.POINTER(C.c_int)(), C.POINTER(C.c_int)(), C.c_int()
    score = L.lmo_chainer(oa, n, 50.0, L.lmo_seed_weight(17.0), 1000.0, 0, C.byref(coff), C.byref(cidx), C.byref(nch))
    chains = [[cidx[j] for j in range(coff[c], coff[c + 1])] for c in range(nch.value)]
    import numpy as np
    opt = O.CmpOpt()
    opt.k, opt.min_prefix = K, 11
    opt.c2.max_gap, opt.c2.min_score, opt.c2.min_align_len = 20, 35, 50
    opt.c2.min_identity, opt.c2.band_count, opt.c2.band_base, opt.c2.heuristic_pident = 70.0, 50, 100, 15.0
    cmp_ = L.lmo_cmp_new(C.byref(opt))
    assert L.lmo_cmp_index(cmp_, s1.encode(), len(s1)) == 0
    ch = C.POINTER(O.Chain2)()
    nc = L.lmo_cmp_compare(cmp_, 0, len(s2) - 1, s2.encode(), len(s2), len(s2), C.byref(ch), None, None)
    cmp_chains = [[ch[i].qbegin, ch[i].qend, ch[i].tbegin, ch[i].tend, ch[i].nanchors, ch[i].matched_bases, ch[i].aligned_bases_q,
                   repr(ch[i].pident)] for i in range(nc)]
    out = dict(source="inputs: /root/reference/lexicmap/cmd/lib-chaining_test.go:33-89, lib-seq_compare_test.go:51-52; outputs: oracle",
               chaining=dict(subs=subs, cleared=cleared, chains=chains, score_f32_bits=int(np.float32(score).view(np.uint32))),
               compare=dict(s1=s1, s2=s2, chains=cmp_chains))
    path = os.path.join(ROOT, "tests", "golden", "ref_test_inputs.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path, "chains:", chains, "compare:", cmp_chains)


if __name__ == "__main__":
    main()
