"""lexicmap_amd/csrc/lm_wfa_mw2_fwd.h (product header: the forward pass of k_wfa_mw2, the workgroup WFA kernel - four
wavefronts per alignment, 256 / 512 / 1024 diagonals; three barriers per score instead of four, ballot trimming, fused
extension, a ring without wrap that the workgroup recentres) - on the host SIMT emulator (tests/emu) against the oracle: score,
run list, coordinates, statistics; wavefronts wider than one wavefront's 64 lanes, than 256 and 512 diagonals (long end gaps),
drifting ones (recentres), the windowed form, and what does not fit says so (status 3)."""
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
        path = os.path.join(EMU, "libwfa_mw2_emu.so")
        srcs = [os.path.join(EMU, f) for f in ("wfa_mw2_emu.cpp", "wfa_host_walk.h", "simt_emu.h")] + [os.path.join(EXP, f) for f in ("lm_wfa_mw2_fwd.h", "lm_wfa_lean2_fwd.h")]
        if not os.path.exists(path) or os.path.getmtime(path) < max(os.path.getmtime(s) for s in srcs):
            subprocess.check_call(["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-o", path, srcs[0]])
        _lib = C.CDLL(path)
        _lib.mw2_emu_run.restype = C.c_long
    return _lib


def run1(q, t, ncw, max_score=20000, arena_cap=1 << 22, win=False):
    cap = len(q) + len(t) + 8
    ops = (C.c_uint64 * cap)()
    o = EmuOut()
    nrec = C.c_int(0)
    n = lib().mw2_emu_run(ncw, int(win), q, len(q), t, len(t), max_score, arena_cap, ops, cap, C.byref(o), C.byref(nrec))
    assert n > 0
    return o.status, (0, o.score, [ops[j] for j in range(o.nops)], o.qbegin, o.qend, o.tbegin, o.tend, o.align_len, o.matches, o.gaps,
                      o.gap_regions), nrec.value


@pytest.mark.parametrize("ncw,n,div,ins,at_end,seed", [(1, 400, 0.08, 0, False, 1), (2, 1500, 0.10, 0, False, 2), (2, 2500, 0.06, 300, False, 3),
                                                       (2, 2500, 0.08, 300, True, 4), (4, 3000, 0.05, 700, True, 5), (2, 900, 0.30, 0, False, 6),
                                                       (4, 2600, 0.10, -600, True, 7)])
def test_workgroup_alignment_equals_the_oracle(ncw, n, div, ins, at_end, seed):
    rng = random.Random(seed)
    q = rand_seq(rng, n)
    if ins >= 0:
        t = with_insertion(rng, q, -1 if at_end else len(q) // 2, ins, div)
    else:
        t = mutate(rng, q, div, div / 4, div / 4)
        q = q + rand_seq(rng, -ins)
    exp = run_oracle_wfa(q, t)
    assert exp[0] == 0
    st, got, _ = run1(q, t, ncw)
    assert st == 0, (st, got[1])
    assert got == exp
    if at_end and div >= 0.08:  # these really are wide: half the ring does not hold them
        st2, got2, _ = run1(q, t, ncw // 2)
        assert st2 == 3 and got2[1] > 256 * (ncw // 2)


def test_drift_recentres_and_statuses():
    rng = random.Random(11)
    q = rand_seq(rng, 6000)
    t = mutate(rng, q, 0.03, 0.0, 0.06)  # one-sided: the final diagonal is ~350 below diagonal 0
    exp = run_oracle_wfa(q, t)
    st, got, nrec = run1(q, t, 2)
    assert st == 0 and got == exp
    for a, b in ((b"ACGT", b"ACGGT"), (b"A", b"A"), (b"ACGTACGTAC", b"TTTTTTTT"), (rand_seq(rng, 33), rand_seq(rng, 31))):
        st, got, _ = run1(a, b, 1)
        assert st in (0, 2) and got == run_oracle_wfa(a, b)
    assert run1(b"ACGTNACGT" * 5, b"ACGTACGT" * 5, 1)[0] == 3
    far = (rand_seq(rng, 120), rand_seq(rng, 120))
    assert run1(far[0], far[1], 1, max_score=40)[0] == 1
    assert run1(far[0], far[1], 1, arena_cap=200)[0] == 1
    wide = rand_seq(rng, 2600)
    tw = with_insertion(rng, wide, -1, 600, 0.12)
    st, got, _ = run1(wide, tw, 2)
    assert st == 3 and got[1] > 512
    assert run1(wide, tw, 4)[:2] == (0, run_oracle_wfa(wide, tw))


@pytest.mark.parametrize("ncw,n,div,ins,seed", [(1, 700, 0.10, 0, 31), (2, 5200, 0.05, 0, 32), (2, 7000, 0.06, 300, 33), (4, 4500, 0.08, 600, 34)])
def test_windowed_form_equals_the_oracle(ncw, n, div, ins, seed):
    rng = random.Random(seed)
    q = rand_seq(rng, n)
    t = with_insertion(rng, q, -1, ins, div)
    exp = run_oracle_wfa(q, t)
    st, got, _ = run1(q, t, ncw, win=True)
    assert st == 0 and got == exp
    if n <= 5200:
        assert run1(q, t, ncw)[:2] == (st, got)
    assert run1(b"ACGTNACGT" * 5, b"ACGTACGT" * 5, 1, win=True)[0] == 3


def test_wandering_wavefront_recentres_the_ring():
    """deletions in the first half, insertions in the second: the best diagonals walk ~150 below diagonal 0 and come back to a
    final diagonal near 0 - the 256-slot frame (centred on 0) has to move with them"""
    rng = random.Random(21)
    q = rand_seq(rng, 6000)
    t = mutate(rng, q[:3000], 0.02, 0.0, 0.055) + mutate(rng, q[3000:], 0.02, 0.05, 0.0)
    assert abs(len(t) - len(q)) < 60
    exp = run_oracle_wfa(q, t)
    st, got, nrec = run1(q, t, 1)
    assert st == 0 and got == exp and nrec >= 1
    st, got, nrec = run1(q, t, 1, win=True)
    assert st == 0 and got == exp and nrec >= 1


def test_the_first_touch_of_a_sequence_end_is_the_end():
    """one edit in the middle of otherwise identical sequences: the score step that leaves the interior mode is also the last"""
    rng = random.Random(71)
    q = rand_seq(rng, 900)
    for t in (q[:450] + (b"A" if q[450:451] != b"A" else b"C") + q[451:], q[:450] + b"ACGTT" + q[450:], q[:450] + q[457:], q):
        exp = run_oracle_wfa(q, t)
        for ncw, win in ((1, False), (2, True)):
            assert run1(q, t, ncw, win=win)[:2] == (0, exp)
