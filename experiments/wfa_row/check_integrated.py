#!/usr/bin/env python3
"""experiments/wfa_row/check_integrated.py - the scratch library with k_wfa_mw wired in (make_integrated.py) against itself with
the switch off: one batch of ONT-style reads of 35-50 kb (HSPs of the 32-65-kb class: 512 / 1024-diagonal passes) searched
with LM_WFA_MW=1 and LM_WFA_MW=0, every column of every row compared, the kernels that ran and their times printed.
    LEXICMAP_HIP_LIB=experiments/lib_mw/liblexicmap_hip.so python experiments/wfa_row/check_integrated.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def ont_read(rng, ref, sub=0.02, ins=0.02, dele=0.03):
    a = np.frombuffer(ref, dtype=np.uint8).copy()
    code = {65: 0, 67: 1, 71: 2, 84: 3}
    q = np.array([code.get(int(x), 0) for x in a], dtype=np.uint8) if len(a) < 1000 else np.select([a == 65, a == 67, a == 71, a == 84], [0, 1, 2, 3], 0).astype(np.uint8)
    m = rng.random(len(q)) < sub
    q[m] = (q[m] + rng.integers(1, 4, int(m.sum()), dtype=np.uint8)) & 3
    q = q[rng.random(len(q)) >= dele]
    k = int(rng.binomial(len(q), ins))
    q = np.insert(q, rng.integers(0, len(q) + 1, k), rng.integers(0, 4, k, dtype=np.uint8))
    return np.frombuffer(b"ACGT", dtype=np.uint8)[q].tobytes()


def run(mw, reads):
    import lexicmap_amd as la
    os.environ["LM_WFA_MW"] = "1" if mw else "0"
    gi = la.Index.synthetic(genomes=12, genome_len=400_000, families=2, seed=77, max_div=0.05, masks=20000)
    if reads is None:
        rng = np.random.default_rng(5)
        reads = []
        for i in range(24):
            L = int(rng.integers(35_000, 50_000))
            reads.append(ont_read(rng, gi.fetch(i % 12, int(rng.integers(0, 400_000 - L)), L)))
    gi.profile(True)
    rows, st = gi.search(reads)
    prof = {p["name"]: (p["launches"], round(p["total_ms"], 2)) for p in gi.profile_get() if p["name"].startswith("k_wfa")}
    gi.close()
    return reads, rows, prof


def main():
    reads, rows1, prof1 = run(True, None)
    _, rows0, prof0 = run(False, reads)
    same = len(rows0) == len(rows1) and all(a == b for a, b in zip(rows0, rows1))
    longest = max((r["aligned_length"] for r in rows1), default=0)
    out = dict(rows=len(rows1), rows_equal=bool(same), longest_hsp=longest, kernels_mw=prof1, kernels_lean=prof0,
               ran_mw=any(k.startswith("k_wfa_mw") for k in prof1))
    print(json.dumps(out))
    return 0 if same and out["ran_mw"] and rows1 else 1


if __name__ == "__main__":
    sys.exit(main())
