"""lexicmap_amd/csrc/lm_pa_chain_pipe_dp.h (product header, switch LM_PA_CHAIN_PIPE): the banded chaining DP of the
pseudo-alignment by a workgroup of 8 wavefronts pipelined over the anchors of ONE window - each wavefront evaluates the band
of its anchor ahead of time (coordinates only), reduces the candidates whose scores are final, then takes the pending ones in
order as the wavefronts behind it publish them - on the host SIMT emulator (tests/emu: 512 fibers, spin-waits yield) against
lm_run_chain2 (the CPU-checked statement of the device logic, itself equal to the oracle's Chainer2): every score and
predecessor, the best score and its anchor."""
import ctypes as C
import os
import random
import subprocess

import pytest

from test_pa_chain_emulated_cpu import colinear

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "emu")
_lib = None


def lib():
    global _lib
    if _lib is None:
        root = os.path.dirname(HERE)
        path = os.path.join(EMU, "libpa_chain_pipe_emu.so")
        srcs = [os.path.join(EMU, "pa_chain_pipe_emu.cpp"), os.path.join(root, "lexicmap_amd", "csrc", "lm_pa_chain_pipe_dp.h"),
                os.path.join(EMU, "simt_emu.h"), os.path.join(root, "lexicmap_amd", "csrc", "lm_algos.h")]
        if not os.path.exists(path) or os.path.getmtime(path) < max(os.path.getmtime(s) for s in srcs):
            subprocess.check_call(["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-o", path, srcs[0]])
        _lib = C.CDLL(path)
    return _lib


def check(anchors, max_gap=20, band_base=100, band_count=50, sched=0):
    n = len(anchors)
    qb = (C.c_int32 * n)(*[a[0] for a in anchors])
    tb = (C.c_int32 * n)(*[a[1] for a in anchors])
    ln = (C.c_uint8 * n)(*[a[2] for a in anchors])
    M, Mi, nc = C.c_longlong(), C.c_int(), C.c_long()
    return lib().pcp_emu_check(qb, tb, ln, n, max_gap, band_base, band_count, C.byref(M), C.byref(Mi), C.byref(nc), C.c_ulonglong(sched)), M.value, Mi.value


@pytest.mark.parametrize("n,seed,kw", [(2, 1, {}), (3, 2, {}), (7, 11, {}), (8, 12, {}), (9, 13, {}), (64, 3, {}), (65, 4, {}), (130, 5, {}),
                                       (700, 6, {}), (2000, 7, {}),
                                       (600, 8, dict(band_base=2000, band_count=200)),   # bands of several candidate rounds
                                       (500, 9, dict(band_base=0, band_count=3)), (400, 10, dict(max_gap=0))])
def test_pipelined_dp_equals_lm_run_chain2(n, seed, kw):
    rng = random.Random(seed)
    for rep in range(2):
        bad, M, Mi = check(colinear(rng, n + rep, step=(1, 8) if kw.get("band_base", 0) > 1000 else (5, 60)), **kw)
        assert bad == 0


def test_dense_and_degenerate_inputs():
    rng = random.Random(99)
    assert check([(10, 10, 20), (10, 50, 20)])[0] == 0                      # same query position: skipped
    assert check([(i, 1000 - i, 15) for i in range(300)])[0] == 0            # anti-diagonal: every candidate is 'after' on t
    assert check([(i * 3, i * 3, 31) for i in range(1000)])[0] == 0          # a perfect diagonal of overlapping anchors
    assert check(sorted((rng.randrange(0, 50), rng.randrange(0, 50), 11) for _ in range(200)))[0] == 0  # everything in one band
    assert check([(5 * i, 7, 12) for i in range(100)])[0] == 0               # one target position: gaps grow, bands empty out


def test_other_interleavings_of_the_wavefronts_give_the_same_scores():
    """the emulator's own schedule is a fixed round robin; with random hand-overs at the pipeline's scheduling points the
    wavefronts overtake each other differently in every run (how far ahead a wavefront evaluates its band, how many candidates
    are still pending when it gets to them): the scores must not depend on it"""
    rng = random.Random(5)
    anchors = colinear(rng, 600)
    dense = colinear(rng, 350, step=(1, 8))
    for seed in range(1, 5):
        assert check(anchors, sched=seed * 7919)[0] == 0
        assert check(dense, band_base=2000, band_count=200, sched=seed * 104729)[0] == 0
