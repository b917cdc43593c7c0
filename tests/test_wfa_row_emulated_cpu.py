"""experiments/wfa_row: the four-alignments-per-wavefront WFA forward pass (measured at 1.12-1.15 x and NOT adopted; kept
as the emulator's first user) run on the host SIMT emulator (tests/emu/simt_emu.h) and checked against the oracle: score, run list, coordinates and statistics of every alignment
that fits its 16*NCR diagonals; what does not fit must say so (status 3), never give a different alignment."""
import ctypes as C
import os
import random
import subprocess

import pytest

import oracle as O
from test_device_algos_cpu import mutate, rand_seq, run_oracle_wfa

HERE = os.path.dirname(os.path.abspath(__file__))
EXP = os.path.join(os.path.dirname(HERE), "experiments", "wfa_row")
EMU = os.path.join(HERE, "emu")   # the host SIMT emulator and the harnesses built on it
MODES = (0, 1, 2)  # WR_EXT_MODE: how the greedy extension loops over the cells of a lane


class EmuOut(C.Structure):
    _fields_ = [("status", C.c_int32), ("score", C.c_int32), ("nops", C.c_int32), ("qbegin", C.c_int32), ("qend", C.c_int32),
                ("tbegin", C.c_int32), ("tend", C.c_int32), ("align_len", C.c_uint32), ("matches", C.c_uint32),
                ("gaps", C.c_uint32), ("gap_regions", C.c_uint32), ("used", C.c_int32)]


_libs = {}


def lib(mode=2):
    if mode not in _libs:
        path = os.path.join(EMU, "libwfa_row_emu_m%d.so" % mode)
        srcs = [os.path.join(EMU, "wfa_row_emu.cpp"), os.path.join(EXP, "wfa_row_fwd.h"), os.path.join(EMU, "simt_emu.h"),
                os.path.join(EMU, "wfa_host_walk.h")]
        if not os.path.exists(path) or os.path.getmtime(path) < max(os.path.getmtime(s) for s in srcs):
            subprocess.check_call(["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-DWR_EXT_MODE=%d" % mode, "-o", path, srcs[0]])
        _libs[mode] = C.CDLL(path)
        _libs[mode].wr_emu_run4.restype = C.c_long
    return _libs[mode]


def run4(pairs, ncr, seq_words=130, max_score=4096, arena_cap=1 << 18, mode=2):
    """one emulated wavefront over up to four (q, t) pairs -> list of (status, tuple comparable with run_oracle_wfa)"""
    L = lib(mode)
    n = len(pairs)
    qs = (C.c_char_p * 4)(*([p[0] for p in pairs] + [b""] * (4 - n)))
    ts = (C.c_char_p * 4)(*([p[1] for p in pairs] + [b""] * (4 - n)))
    ql = (C.c_int32 * 4)(*([len(p[0]) for p in pairs] + [0] * (4 - n)))
    tl = (C.c_int32 * 4)(*([len(p[1]) for p in pairs] + [0] * (4 - n)))
    cap = max(len(p[0]) + len(p[1]) for p in pairs) + 8
    bufs = [(C.c_uint64 * cap)() for _ in range(4)]
    ops = (C.POINTER(C.c_uint64) * 4)(*[C.cast(b, C.POINTER(C.c_uint64)) for b in bufs])
    out = (EmuOut * 4)()
    nc = L.wr_emu_run4(ncr, qs, ql, ts, tl, n, seq_words, max_score, arena_cap, ops, cap, out)
    assert nc > 0
    res = []
    for i in range(n):
        o = out[i]
        res.append((o.status, (0, o.score, [bufs[i][j] for j in range(o.nops)], o.qbegin, o.qend, o.tbegin, o.tend, o.align_len,
                               o.matches, o.gaps, o.gap_regions)))
    return res


def gene_pairs(rng, n, lo, hi, div):
    out = []
    for _ in range(n):
        q = rand_seq(rng, rng.randrange(lo, hi))
        d = rng.random() * div
        out.append((q, mutate(rng, q, d, d / 4, d / 4)))
    return out


@pytest.mark.parametrize("ncr,lo,hi,div,seed,mode", [(4, 30, 400, 0.10, 1, 0), (4, 300, 1500, 0.06, 2, 2), (8, 300, 1800, 0.15, 3, 1),
                                                     (2, 20, 200, 0.05, 4, 2), (8, 1000, 2000, 0.30, 5, 2), (4, 300, 1500, 0.08, 6, 1)])
def test_rows_equal_the_oracle(ncr, lo, hi, div, seed, mode):
    rng = random.Random(seed)
    pairs = gene_pairs(rng, 16, lo, hi, div)
    n_ok = 0
    for g in range(0, len(pairs), 4):
        got = run4(pairs[g:g + 4], ncr, mode=mode)
        for (q, t), (st, tup) in zip(pairs[g:g + 4], got):
            exp = run_oracle_wfa(q, t)
            assert exp[0] == 0
            assert st in (0, 2, 3), st
            if st == 3:  # wider than 16*ncr - 2 diagonals at some score: the next ring width takes it
                assert tup[1] > 16 * ncr - 2
                continue
            assert tup == exp, (len(q), len(t))
            n_ok += 1
    assert n_ok >= 8  # most gene-sized alignments fit


@pytest.mark.parametrize("mode", MODES)
def test_rows_of_one_wavefront_are_independent(mode):
    """very different lengths and divergences side by side, fewer than four problems, a problem alone: same results"""
    rng = random.Random(9)
    a = gene_pairs(rng, 1, 1200, 1300, 0.08)[0]
    b = gene_pairs(rng, 1, 40, 60, 0.0)[0]
    c = gene_pairs(rng, 1, 500, 600, 0.2)[0]
    d = (rand_seq(rng, 35), rand_seq(rng, 33))  # unrelated: a high score on a short pair
    alone = [run4([p], 8, mode=mode)[0] for p in (a, b, c, d)]
    for order in ([a, b, c, d], [d, c, b, a], [b, a], [c, d, a]):
        got = run4(order, 8, mode=mode)
        for p, g in zip(order, got):
            assert g == alone[(a, b, c, d).index(p)]
    for p, g in zip((a, b, c, d), alone):
        if g[0] != 3:
            assert g[1] == run_oracle_wfa(*p)


def test_what_does_not_fit_says_so():
    rng = random.Random(11)
    q = rand_seq(rng, 300)
    # a 40-base insertion needs 40+ diagonals: too wide for 32 (ncr 2), fine for 128
    t = q[:150] + rand_seq(rng, 40) + q[150:]
    assert run4([(q, t)], 2)[0][0] == 3
    assert run4([(q, t)], 8)[0] == (0, run_oracle_wfa(q, t))
    # not plain ACGT, longer than the LDS sequence buffers, score beyond the header: status 3 / 3 / 1, neighbours unaffected
    ok = gene_pairs(rng, 1, 200, 300, 0.05)[0]
    got = run4([(b"ACGTNACGT" * 5, b"ACGTACGT" * 5), ok, (rand_seq(rng, 130 * 16 + 5), rand_seq(rng, 50))], 4)
    assert got[0][0] == 3 and got[2][0] == 3
    assert got[1] == (0, run_oracle_wfa(*ok))
    far = (rand_seq(rng, 120), rand_seq(rng, 120))
    got = run4([far, ok], 8, max_score=40)
    assert got[0][0] == 1 and got[1] == (0, run_oracle_wfa(*ok))
    got = run4([far, ok], 8, arena_cap=200)
    assert got[0][0] == 1


def test_the_emulator_itself(tmp_path):
    """cross-lane results are right, and lanes that do not meet at the same operation stop the run (SIGABRT) instead of
    returning a value - the property the CPU check of a kernel rests on"""
    exe = str(tmp_path / "simt_emu_selftest")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-o", exe, os.path.join(EMU, "simt_emu_selftest.cpp")])
    assert subprocess.run([exe], capture_output=True, text=True).stdout.startswith("ok")
    for how in ("diverge", "early", "stuck"):
        r = subprocess.run([exe, how], capture_output=True, text=True)
        assert r.returncode == -6 and "simt_emu:" in r.stderr
