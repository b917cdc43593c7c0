#!/usr/bin/env python3
"""experiments/wfa_row/probe.py - GPU probe of the staged four-alignments-per-wavefront WFA kernel against the product's
k_wfa_lean on the same synthetic HSP pairs (gene-sized by default): parity (every record and operation) and kernel time.
    python experiments/wfa_row/probe.py [--n 65536] [--len 1000 2000] [--div 0.10] [--ncr 4 8] [--reps 3]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class Cmp(C.Structure):
    _fields_ = [("ms_lean", C.c_double), ("ms_row", C.c_double), ("n", C.c_int64), ("lean_ok", C.c_int64), ("row_ok", C.c_int64),
                ("row_status3", C.c_int64), ("row_status1", C.c_int64), ("lean_status3", C.c_int64), ("both_ok", C.c_int64),
                ("mismatches", C.c_int64), ("blocks_lean", C.c_int32), ("blocks_row", C.c_int32), ("order_by_score", C.c_int32),
                ("ms_lean_left", C.c_double), ("n_routed", C.c_int64)]


def make_pairs(n, lo, hi, div, seed):
    """n (query, target) pairs like the HSP windows WFA sees: target = query with substitutions at rate d and insertions /
    deletions at d/4 each, d uniform in [0, div]"""
    rng = np.random.default_rng(seed)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    chunks, qoff, qlen, toff, tlen = [], [], [], [], []
    pos = 0
    for _ in range(n):
        L = int(rng.integers(lo, hi + 1))
        d = float(rng.random()) * div
        q = rng.integers(0, 4, L, dtype=np.uint8)
        t = q.copy()
        sub = rng.random(L) < d
        t[sub] = (t[sub] + rng.integers(1, 4, int(sub.sum()), dtype=np.uint8)) & 3
        keep = rng.random(L) >= d / 4
        t = t[keep]
        nins = int(rng.binomial(L, d / 4))
        if nins:
            t = np.insert(t, rng.integers(0, len(t) + 1, nins), rng.integers(0, 4, nins, dtype=np.uint8))
        qa, ta = alpha[q], alpha[t]
        chunks += [qa, ta]
        qoff.append(pos)
        qlen.append(len(qa))
        pos += len(qa)
        toff.append(pos)
        tlen.append(len(ta))
        pos += len(ta)
    return np.concatenate(chunks), np.array(qoff, np.int64), np.array(qlen, np.int32), np.array(toff, np.int64), np.array(tlen, np.int32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=65536)
    ap.add_argument("--len", type=int, nargs=2, default=[1000, 2000])
    ap.add_argument("--div", type=float, default=0.10)
    ap.add_argument("--ncr", type=int, nargs="+", default=[4, 8])
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--by-score", action="store_true", help="queue in decreasing order of the true score (the product orders by expected cost)")
    ap.add_argument("--route", type=int, nargs="+", default=[0], help="score per 1000 bases of q+t above which a problem skips the row kernel")
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--libs", nargs="+", default=["libwfa_row_exp.so"])
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    t0 = time.time()
    seqs, qoff, qlen, toff, tlen = make_pairs(a.n, a.len[0], a.len[1], a.div, a.seed)
    gen_s = time.time() - t0
    out = []
    for libname, ncr, thr in [(x, y, z) for x in a.libs for y in a.ncr for z in a.route]:
        L = C.CDLL(os.path.join(HERE, libname))
        r = Cmp()
        st = L.wr_compare(seqs.ctypes.data_as(C.c_void_p), C.c_int64(len(seqs)), qoff.ctypes.data_as(C.c_void_p), qlen.ctypes.data_as(C.c_void_p),
                          toff.ctypes.data_as(C.c_void_p), tlen.ctypes.data_as(C.c_void_p), C.c_int64(a.n), ncr, a.reps, int(a.by_score), thr, C.byref(r))
        d = {f[0]: getattr(r, f[0]) for f in Cmp._fields_ if f[0] != "pad"}
        d.update(lib=libname, rc=st, ncr=ncr, route_permille=thr, diagonals_per_alignment=16 * ncr, pairs=a.n, length=a.len, max_div=a.div,
                 speedup=(r.ms_lean / r.ms_row if r.ms_row > 0 else None),
                 speedup_with_leftovers=(r.ms_lean / (r.ms_row + r.ms_lean_left) if r.ms_row > 0 else None),
                 row_share_aligned=(r.row_ok / a.n if a.n else None))
        out.append(d)
        print(json.dumps(d), flush=True)
    if a.out:
        json.dump(dict(generated_s=round(gen_s, 1), runs=out), open(a.out, "w"), indent=1)
    return 0 if all(d["rc"] == 0 and d["mismatches"] == 0 for d in out) else 1


if __name__ == "__main__":
    sys.exit(main())
