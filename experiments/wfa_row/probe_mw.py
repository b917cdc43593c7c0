#!/usr/bin/env python3
"""experiments/wfa_row/probe_mw.py - GPU probe of the staged workgroup-per-alignment WFA kernel (k_wfa_mw<2> / <4>: four
wavefronts, 512 / 1024 diagonals) against the product's single-wavefront k_wfa_lean<8> / <16> on the same long ONT-style
pairs - the handful of 20-45-kb alignments every round of the C3 pipeline waits for: parity and kernel time.
    python experiments/wfa_row/probe_mw.py [--n 48] [--len 20000 45000] [--ncw 2 4]"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class Cmp(C.Structure):
    _fields_ = [("ms_lean", C.c_double), ("ms_mw", C.c_double), ("n", C.c_int64), ("lean_ok", C.c_int64), ("mw_ok", C.c_int64),
                ("lean_status3", C.c_int64), ("mw_status3", C.c_int64), ("mw_status1", C.c_int64), ("both_ok", C.c_int64),
                ("mismatches", C.c_int64), ("blocks_lean", C.c_int32), ("blocks_mw", C.c_int32), ("max_width_reported", C.c_int32),
                ("pad", C.c_int32)]


def ont_pairs(n, lo, hi, seed, sub=0.02, ins=0.02, dele=0.03):
    """(read, reference window) pairs: the read = the window with substitutions, insertions and deletions at ONT-like rates
    (more deletions than insertions: the final diagonal drifts ~1 % of the length away from the first one)"""
    rng = np.random.default_rng(seed)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    chunks, qoff, qlen, toff, tlen = [], [], [], [], []
    pos = 0
    for _ in range(n):
        L = int(rng.integers(lo, hi + 1))
        t = rng.integers(0, 4, L, dtype=np.uint8)
        q = t.copy()
        m = rng.random(L) < sub
        q[m] = (q[m] + rng.integers(1, 4, int(m.sum()), dtype=np.uint8)) & 3
        q = q[rng.random(L) >= dele]
        k = int(rng.binomial(L, ins))
        q = np.insert(q, rng.integers(0, len(q) + 1, k), rng.integers(0, 4, k, dtype=np.uint8))
        qa, ta = alpha[q], alpha[t]
        chunks += [qa, ta]
        qoff.append(pos); qlen.append(len(qa)); pos += len(qa)
        toff.append(pos); tlen.append(len(ta)); pos += len(ta)
    return np.concatenate(chunks), np.array(qoff, np.int64), np.array(qlen, np.int32), np.array(toff, np.int64), np.array(tlen, np.int32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=48)
    ap.add_argument("--len", type=int, nargs=2, default=[20000, 45000])
    ap.add_argument("--ncw", type=int, nargs="+", default=[2, 4])
    ap.add_argument("--win", type=int, nargs="+", default=[0], help="1: the windowed form (against k_wfa_lean<nc, true>)")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--seed", type=int, default=11)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    L = C.CDLL(os.path.join(HERE, "libwfa_mw_exp.so"))
    seqs, qoff, qlen, toff, tlen = ont_pairs(a.n, a.len[0], a.len[1], a.seed)
    out = []
    for ncw, win in [(x, y) for x in a.ncw for y in a.win]:
        r = Cmp()
        st = L.mw_compare(seqs.ctypes.data_as(C.c_void_p), C.c_int64(len(seqs)), qoff.ctypes.data_as(C.c_void_p), qlen.ctypes.data_as(C.c_void_p),
                          toff.ctypes.data_as(C.c_void_p), tlen.ctypes.data_as(C.c_void_p), C.c_int64(a.n), ncw, win, a.reps, C.byref(r))
        d = {f[0]: getattr(r, f[0]) for f in Cmp._fields_ if f[0] != "pad"}
        d.update(rc=st, ncw=ncw, diagonals=256 * ncw, windowed=bool(win), against="k_wfa_lean<%d, %s>" % (4 * ncw, "true" if win else "false"), pairs=a.n, length=a.len,
                 drift=[int(x) for x in np.sort(tlen - qlen)[[0, len(tlen) // 2, -1]]],
                 speedup=(r.ms_lean / r.ms_mw if r.ms_mw > 0 else None))
        out.append(d)
        print(json.dumps(d), flush=True)
    if a.out:
        json.dump(dict(runs=out), open(a.out, "w"), indent=1)
    return 0 if all(d["rc"] == 0 and d["mismatches"] == 0 for d in out) else 1


if __name__ == "__main__":
    sys.exit(main())
