"""lexicmap_amd/csrc/lm_wfa_mw_fwd.h - the forward pass of the product's k_wfa_mw (a workgroup of four wavefronts per long
alignment; one source for the device and for the host) - on the host SIMT emulator (tests/emu) against the oracle: score, run list, coordinates and statistics;
wavefronts wider than one wavefront's 64 lanes, than 256 and than 512 diagonals (long insertions), and what does not fit
256 * NCW - 2 diagonals must say so (status 3)."""
import ctypes as C
import os
import random
import subprocess

import pytest

from test_device_algos_cpu import mutate, rand_seq, run_oracle_wfa
from test_wfa_row_emulated_cpu import EMU, EmuOut

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(EMU, "libwfa_mw_emu.so")
        srcs = [os.path.join(EMU, f) for f in ("wfa_mw_emu.cpp", "wfa_host_walk.h", "simt_emu.h")]
        srcs.append(os.path.join(os.path.dirname(os.path.dirname(EMU)), "lexicmap_amd", "csrc", "lm_wfa_mw_fwd.h"))
        if not os.path.exists(path) or os.path.getmtime(path) < max(os.path.getmtime(s) for s in srcs):
            subprocess.check_call(["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-o", path, srcs[0]])
        _lib = C.CDLL(path)
        _lib.mw_emu_run.restype = C.c_long
    return _lib


def run1(q, t, ncw, seq_words=None, max_score=20000, arena_cap=1 << 22, win=False):
    L = lib()
    if seq_words is None:
        seq_words = (max(len(q), len(t)) + 15) // 16 + 1
    cap = len(q) + len(t) + 8
    ops = (C.c_uint64 * cap)()
    o = EmuOut()
    n = L.mw_emu_run(ncw, int(win), q, len(q), t, len(t), seq_words, max_score, arena_cap, ops, cap, C.byref(o))
    assert n > 0
    return o.status, (0, o.score, [ops[j] for j in range(o.nops)], o.qbegin, o.qend, o.tbegin, o.tend, o.align_len, o.matches, o.gaps,
                      o.gap_regions)


def with_insertion(rng, q, at, n, div):
    """target = mutated query with n extra bases at `at` (-1: at the END - the final diagonal n is then never trimmed away
    (lm_wfa_align keeps the range open towards it), so the wavefront grows one diagonal per score all along the alignment
    and ends up |n| + ~50 wide: the way the 512 / 1024-diagonal rings get used)"""
    t = mutate(rng, q, div, div / 4, div / 4)
    return t + rand_seq(rng, n) if at < 0 else t[:at] + rand_seq(rng, n) + t[at:]


@pytest.mark.parametrize("ncw,n,div,ins,at_end,seed", [(1, 400, 0.08, 0, False, 1), (2, 1500, 0.10, 0, False, 2), (2, 2500, 0.06, 300, False, 3),
                                                       (2, 2500, 0.08, 300, True, 4), (4, 3000, 0.05, 700, True, 5), (2, 900, 0.30, 0, False, 6),
                                                       (4, 2600, 0.10, -600, True, 7), (4, 3000, 0.10, 700, True, 8)])
def test_workgroup_alignment_equals_the_oracle(ncw, n, div, ins, at_end, seed):
    rng = random.Random(seed)
    for rep in range(2):
        q = rand_seq(rng, n + 17 * rep)
        if ins >= 0:
            t = with_insertion(rng, q, -1 if at_end else len(q) // 2, ins, div)
        else:  # the query has the extra bases (at its end): the final diagonal is negative
            t = mutate(rng, q, div, div / 4, div / 4)
            q = q + rand_seq(rng, -ins)
        exp = run_oracle_wfa(q, t)
        assert exp[0] == 0
        st, got = run1(q, t, ncw)
        assert st == 0, (st, got[1])
        assert got == exp
        if at_end and div >= 0.08:  # these really are wide (enough score steps before the end gap): half the ring does not hold them
            st2, got2 = run1(q, t, ncw // 2)
            assert st2 == 3 and got2[1] > 256 * (ncw // 2) - 2


@pytest.mark.parametrize("ncw,n,div,ins,seed", [(1, 700, 0.10, 0, 31), (2, 5200, 0.05, 0, 32), (2, 9000, 0.06, 300, 33), (4, 6000, 0.10, 600, 34)])
def test_windowed_form_equals_the_oracle(ncw, n, div, ins, seed):
    """WIN: sequences through sliding 4096-base windows (any length; beyond 4096 bases the windows have to move, and with a
    long end gap the cells of one wavefront wait for each other's window positions)"""
    rng = random.Random(seed)
    q = rand_seq(rng, n)
    t = with_insertion(rng, q, -1, ins, div)
    exp = run_oracle_wfa(q, t)
    st, got = run1(q, t, ncw, win=True, seq_words=1)  # seq_words is not used by the windowed form
    assert st == 0 and got == exp
    if n <= 5200:  # and the whole-sequence form of the same kernel agrees
        assert run1(q, t, ncw) == (st, got)


def test_windowed_form_statuses():
    rng = random.Random(41)
    assert run1(b"ACGTNACGT" * 5, b"ACGTACGT" * 5, 1, win=True)[0] == 3          # not plain ACGT
    q = rand_seq(rng, 5000)
    assert run1(q[:2500] + b"N" + q[2500:], q, 1, win=True)[0] == 3               # ... met only after a window move
    far = (rand_seq(rng, 120), rand_seq(rng, 120))
    assert run1(far[0], far[1], 1, max_score=40, win=True)[0] == 1
    st, got = run1(b"ACGT", b"ACGGT", 1, win=True)
    assert (st, got) == (0, run_oracle_wfa(b"ACGT", b"ACGGT"))


def test_too_wide_for_the_ring_says_so_and_small_cases():
    rng = random.Random(21)
    q = rand_seq(rng, 2600)
    t = with_insertion(rng, q, -1, 600, 0.12)  # ~700 score steps, then 600 extra bases at the end: 600+ diagonals
    st, got = run1(q, t, 2)
    assert st == 3 and got[1] > 510
    assert run1(q, t, 4) == (0, run_oracle_wfa(q, t))
    for a, b in ((b"ACGT", b"ACGGT"), (b"A", b"A"), (b"ACGTACGTAC", b"TTTTTTTT"), (rand_seq(rng, 33), rand_seq(rng, 31))):
        st, got = run1(a, b, 1)
        assert st in (0, 2)
        exp = run_oracle_wfa(a, b)
        assert got == exp
    # not plain ACGT, longer than the LDS buffers, score beyond the header, scratch too small
    assert run1(b"ACGTNACGT" * 5, b"ACGTACGT" * 5, 1)[0] == 3
    assert run1(rand_seq(rng, 400), rand_seq(rng, 50), 1, seq_words=20)[0] == 3
    far = (rand_seq(rng, 120), rand_seq(rng, 120))
    assert run1(far[0], far[1], 1, max_score=40)[0] == 1
    assert run1(far[0], far[1], 1, arena_cap=200)[0] == 1
