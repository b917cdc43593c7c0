"""lexicmap_amd/csrc/lm_wfa_lean2_fwd.h (product header: the forward pass of k_wfa_lean2, the single-wavefront WFA kernel - no-wrap ring that is
recentred, trimming by ballots, extension fused behind the recurrence; fewer instructions per score step) - on the host SIMT
emulator (tests/emu) against the oracle: score, run list, coordinates and statistics; rings of 64-512 diagonals with 32- and
16-bit cells; wavefronts that drift (the ring is recentred), that outgrow the ring (status 3) and the small cases."""
import ctypes as C
import os
import random
import subprocess

import pytest

from test_device_algos_cpu import mutate, rand_seq, run_oracle_wfa
from emu_common import EMU, EmuOut, with_insertion

EXP = os.path.join(os.path.dirname(os.path.dirname(EMU)), "lexicmap_amd", "csrc")
_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(EMU, "libwfa_lean2_emu.so")
        srcs = [os.path.join(EMU, f) for f in ("wfa_lean2_emu.cpp", "wfa_host_walk.h", "simt_emu.h")] + [os.path.join(EXP, "lm_wfa_lean2_fwd.h")]
        if not os.path.exists(path) or os.path.getmtime(path) < max(os.path.getmtime(s) for s in srcs):
            subprocess.check_call(["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-o", path, srcs[0]])
        _lib = C.CDLL(path)
        _lib.l2_emu_run.restype = C.c_long
    return _lib


def run1(q, t, nc, r16=False, max_score=20000, arena_cap=1 << 22, win=False):
    """-> status, tuple comparable with run_oracle_wfa, number of times the ring was recentred"""
    cap = len(q) + len(t) + 8
    ops = (C.c_uint64 * cap)()
    o = EmuOut()
    nrec = C.c_int(0)
    n = lib().l2_emu_run(nc, int(r16), int(win), q, len(q), t, len(t), max_score, arena_cap, ops, cap, C.byref(o), C.byref(nrec))
    assert n > 0
    return o.status, (0, o.score, [ops[j] for j in range(o.nops)], o.qbegin, o.qend, o.tbegin, o.tend, o.align_len, o.matches, o.gaps,
                      o.gap_regions), nrec.value


@pytest.mark.parametrize("nc,r16,n,div,seed", [(1, False, 300, 0.05, 1), (1, True, 500, 0.08, 2), (2, True, 1500, 0.10, 3), (2, False, 1500, 0.12, 4),
                                               (4, True, 2500, 0.15, 5), (2, True, 900, 0.30, 6), (4, False, 1200, 0.25, 7), (8, False, 1500, 0.10, 8)])
def test_alignment_equals_the_oracle(nc, r16, n, div, seed):
    rng = random.Random(seed)
    for rep in range(3):
        q = rand_seq(rng, n + 17 * rep)
        t = mutate(rng, q, div, div / 4, div / 4)
        exp = run_oracle_wfa(q, t)
        assert exp[0] == 0
        st, got, _ = run1(q, t, nc, r16)
        if st == 3:  # outgrew the ring: the next width must take it (and say the same as the oracle)
            assert nc < 8
            st, got, _ = run1(q, t, nc * 2, r16 and nc * 2 <= 4)
        assert st == 0, (st, got[1])
        assert got == exp


def drifting(rng, n, sub, ins, dele):
    q = rand_seq(rng, n)
    return q, mutate(rng, q, sub, ins, dele)


@pytest.mark.parametrize("nc,r16,n,sub,ins,dele,seed", [(1, True, 3000, 0.02, 0.0, 0.05, 11), (1, False, 3000, 0.02, 0.05, 0.0, 12),
                                                        (2, True, 6000, 0.03, 0.0, 0.06, 13), (2, True, 6000, 0.03, 0.06, 0.005, 14),
                                                        (4, False, 5000, 0.05, 0.01, 0.08, 15)])
def test_drifting_wavefronts_recentre_the_ring(nc, r16, n, sub, ins, dele, seed):
    """one-sided indels: the final diagonal is 150-400 away from diagonal 0, the live rows leave the frame again and again"""
    rng = random.Random(seed)
    q, t = drifting(rng, n, sub, ins, dele)
    assert abs(len(t) - len(q)) > 64 * nc  # further than the ring is wide
    exp = run_oracle_wfa(q, t)
    st, got, nrec = run1(q, t, nc, r16)
    while st == 3:
        nc *= 2
        st, got, nrec = run1(q, t, nc, r16 and nc <= 4)
    assert st == 0 and got == exp
    assert nrec >= 2


def test_end_gaps_wide_wavefronts_and_what_does_not_fit():
    rng = random.Random(21)
    q = rand_seq(rng, 1500)
    t = with_insertion(rng, q, -1, 150, 0.10)  # the final diagonal is never trimmed away: 150+ diagonals wide at the end
    exp = run_oracle_wfa(q, t)
    st, got, _ = run1(q, t, 2, True)
    assert st == 3 and got[1] > 128
    assert run1(q, t, 4, True)[:2] == (0, exp)
    assert run1(q, t, 8)[:2] == (0, exp)
    # the query has the extra bases: negative final diagonal
    t2 = mutate(rng, q, 0.08, 0.02, 0.02)
    q2 = q + rand_seq(rng, 100)
    assert run1(q2, t2, 4)[:2] == (0, run_oracle_wfa(q2, t2))
    # an insertion in the middle
    t3 = with_insertion(rng, q, 700, 90, 0.05)
    assert run1(q, t3, 4, True)[:2] == (0, run_oracle_wfa(q, t3))


def test_small_cases_and_statuses():
    rng = random.Random(31)
    for a, b in ((b"ACGT", b"ACGGT"), (b"A", b"A"), (b"A", b"C"), (b"ACGTACGTAC", b"TTTTTTTT"), (rand_seq(rng, 33), rand_seq(rng, 31)),
                 (b"ACGTACGTACGTACGTACGT", b"ACGTACGTACGTACGTACGT"), (rand_seq(rng, 16), rand_seq(rng, 48))):
        for nc, r16 in ((1, False), (2, True), (4, True)):
            st, got, _ = run1(a, b, nc, r16)
            assert st in (0, 2)
            assert got == run_oracle_wfa(a, b), (a, b, nc)
    assert run1(b"ACGTNACGT" * 5, b"ACGTACGT" * 5, 1)[0] == 3  # not plain ACGT
    far = (rand_seq(rng, 120), rand_seq(rng, 120))
    assert run1(far[0], far[1], 2, max_score=40)[0] == 1       # score beyond the header
    assert run1(far[0], far[1], 2, arena_cap=200)[0] == 1      # scratch too small
    long = rand_seq(rng, 12500)
    assert run1(long, long, 2, True)[0] == 3                   # 16-bit cells: sequences up to 12 000 bases
    assert run1(long, long, 2, False)[:2] == (0, run_oracle_wfa(long, long))


def test_many_random_pairs():
    """a sweep over short pairs of every shape: lengths 1-400, divergence 0-40 %, unrelated pairs, length differences"""
    rng = random.Random(41)
    n = 0
    for rep in range(120):
        la = rng.randint(1, 400)
        a = rand_seq(rng, la)
        kind = rng.randint(0, 3)
        if kind == 0:
            b = rand_seq(rng, rng.randint(1, 400))
        else:
            d = rng.choice((0.0, 0.02, 0.1, 0.25, 0.4))
            b = mutate(rng, a, d, d / 3, d / 3)
            if kind == 2:
                b = b + rand_seq(rng, rng.randint(1, 60))
            if kind == 3 and len(b) > 40:
                b = b[rng.randint(1, 30):]
        if not b:
            b = b"A"
        exp = run_oracle_wfa(a, b)
        nc, r16 = rng.choice(((1, True), (2, True), (2, False), (4, True)))
        st, got, _ = run1(a, b, nc, r16, max_score=4096, arena_cap=1 << 20)
        while st == 3 and nc < 8:
            nc *= 2
            st, got, _ = run1(a, b, nc, r16 and nc <= 4, max_score=4096, arena_cap=1 << 20)
        if exp[0] == 0:
            assert st == 0 and got == exp, (rep, la, len(b), nc)
            n += 1
        else:
            assert st != 0 or got == exp
    assert n > 80


@pytest.mark.parametrize("nc,n,div,ins,seed", [(1, 700, 0.10, 0, 51), (2, 5200, 0.05, 0, 52), (2, 9000, 0.06, 100, 53), (4, 6000, 0.10, 200, 54),
                                               (8, 5000, 0.08, 400, 55)])
def test_windowed_form_equals_the_oracle(nc, n, div, ins, seed):
    """WIN: the sequences through sliding 4096-base windows filled by the pass itself (any length; beyond 4096 bases the windows
    move, and with a long end gap the cells of one wavefront wait for each other's window positions)"""
    rng = random.Random(seed)
    q = rand_seq(rng, n)
    t = with_insertion(rng, q, -1, ins, div)
    exp = run_oracle_wfa(q, t)
    st, got, _ = run1(q, t, nc, win=True)
    if st == 3:  # (64 diagonals do not hold a 10 % pair: the next width takes it)
        assert nc == 1 and got[1] > 64
        nc = 2
        st, got, _ = run1(q, t, nc, win=True)
    assert st == 0 and got == exp
    if n <= 5200:  # and the whole-sequence form agrees
        assert run1(q, t, nc)[:2] == (st, got)


def test_windowed_form_statuses_and_drift():
    rng = random.Random(61)
    assert run1(b"ACGTNACGT" * 5, b"ACGTACGT" * 5, 1, win=True)[0] == 3          # not plain ACGT
    q = rand_seq(rng, 5000)
    assert run1(q[:2500] + b"N" + q[2500:], q, 1, win=True)[0] == 3               # ... met only after a window move
    far = (rand_seq(rng, 120), rand_seq(rng, 120))
    assert run1(far[0], far[1], 1, max_score=40, win=True)[0] == 1
    assert run1(b"ACGT", b"ACGGT", 1, win=True)[:2] == (0, run_oracle_wfa(b"ACGT", b"ACGGT"))
    q, t = drifting(rng, 7000, 0.03, 0.0, 0.05)  # windows move AND the ring is recentred
    nc = 2
    st, got, nrec = run1(q, t, nc, win=True)
    while st == 3 and nc < 8:  # (the final diagonal is ~350 below diagonal 0 and the cut-off keeps the range open towards it)
        nc *= 2
        st, got, nrec = run1(q, t, nc, win=True)
    assert st == 0 and got == run_oracle_wfa(q, t) and nrec >= 2


def test_the_first_touch_of_a_sequence_end_is_the_end():
    """one edit in the middle of otherwise identical sequences: the extension that follows it runs to the end - the score step
    that leaves the interior mode is also the last one"""
    rng = random.Random(71)
    q = rand_seq(rng, 900)
    for t in (q[:450] + (b"A" if q[450:451] != b"A" else b"C") + q[451:], q[:450] + b"ACGTT" + q[450:], q[:450] + q[457:], q):
        exp = run_oracle_wfa(q, t)
        for nc, r16, win in ((1, False, False), (2, True, False), (2, False, True), (4, True, False)):
            assert run1(q, t, nc, r16, win=win)[:2] == (0, exp)


@pytest.mark.parametrize("nc,win,n,extra,seed", [(8, False, 3000, 330, 41), (8, True, 5000, -400, 42), (16, False, 2500, 700, 43), (16, True, 6000, -820, 44),
                                                  (8, False, 2500, 0, 45), (16, True, 2200, 30, 46)])
def test_wide_rings_run_in_their_flavours(nc, win, n, extra, seed):
    """512 / 1024 diagonals by one wavefront (round 6: the workgroup kernels are gone).  The hot loop exists per flavour - chunks
    0 .. NA-1 of the frame, NA in {2, 4, 8} / {4, 8, 16} - and the rows move between them as the wavefront widens towards a far
    final diagonal (|tlen - qlen| of 330-820) and narrows again behind it; narrow pairs stay in the smallest flavour."""
    rng = random.Random(seed)
    q = rand_seq(rng, n)
    t = mutate(rng, q, 0.03, 0.01, 0.01)
    if extra > 0:
        t = t + rand_seq(rng, extra)
    elif extra < 0:
        q = q + rand_seq(rng, -extra)
    exp = run_oracle_wfa(q, t)
    assert exp[0] == 0
    st, got, nrec = run1(q, t, nc, False, win=win, max_score=40000)
    assert st == 0 and got == exp
    if abs(extra) > 128 * (nc // 8):
        assert nrec >= 1   # the frame was re-cut at least once on the way
