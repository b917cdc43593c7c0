"""lexicmap_amd/csrc/lm_pa_chain_dp_core.h - the banded chaining DP of k_pa_chain_wave with the last 64 anchors in registers and the LDS
ring behind them (one source for the device and for the host) - on the host SIMT emulator (tests/emu) against lm_run_chain2 (the CPU-checked statement of the
device logic, itself equal to the oracle's Chainer2): every score and predecessor, the best score and its anchor."""
import ctypes as C
import os
import random
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "emu")
_lib = None


def lib():
    global _lib
    if _lib is None:
        csrc = os.path.join(os.path.dirname(HERE), "lexicmap_amd", "csrc")
        path = os.path.join(EMU, "libpa_chain_emu.so")
        srcs = [os.path.join(EMU, "pa_chain_emu.cpp"), os.path.join(csrc, "lm_pa_chain_dp_core.h"), os.path.join(EMU, "simt_emu.h"),
                os.path.join(csrc, "lm_algos.h")]
        if not os.path.exists(path) or os.path.getmtime(path) < max(os.path.getmtime(s) for s in srcs):
            subprocess.check_call(["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-o", path, srcs[0]])
        _lib = C.CDLL(path)
    return _lib


def check(anchors, max_gap=20, band_base=100, band_count=50):
    n = len(anchors)
    qb = (C.c_int32 * n)(*[a[0] for a in anchors])
    tb = (C.c_int32 * n)(*[a[1] for a in anchors])
    ln = (C.c_uint8 * n)(*[a[2] for a in anchors])
    M, Mi = C.c_longlong(), C.c_int()
    return lib().pcd_emu_check(qb, tb, ln, n, max_gap, band_base, band_count, None, C.byref(M), C.byref(Mi)), M.value, Mi.value


def colinear(rng, n, step=(5, 60), noise=0.2, jump=0.02):
    """anchors of a pseudo-alignment window, sorted by query position: mostly colinear, some off-diagonal noise, some indel
    jumps, some repeated query positions (skipped by the DP)"""
    out, q, d = [], 0, 0
    for _ in range(n):
        if rng.random() > 0.1:
            q += rng.randrange(*step)
        if rng.random() < jump:
            d += rng.randrange(-30, 31)
        t = q + d + (rng.randrange(-400, 400) if rng.random() < noise else 0)
        out.append((q, max(0, t), rng.randrange(11, 32)))
    out.sort(key=lambda a: (a[0], a[1]))
    return out


@pytest.mark.parametrize("n,seed,kw", [(2, 1, {}), (3, 2, {}), (64, 3, {}), (65, 4, {}), (130, 5, {}), (700, 6, {}), (3000, 7, {}),
                                       (900, 8, dict(band_base=2000, band_count=200)),   # the band reaches behind the ring
                                       (500, 9, dict(band_base=0, band_count=3)), (400, 10, dict(max_gap=0))])
def test_ring_dp_equals_lm_run_chain2(n, seed, kw):
    rng = random.Random(seed)
    for rep in range(3):
        bad, M, Mi = check(colinear(rng, n + rep, step=(1, 8) if kw.get("band_base", 0) > 1000 else (5, 60)), **kw)
        assert bad == 0


def test_dense_and_degenerate_inputs():
    rng = random.Random(99)
    assert check([(10, 10, 20), (10, 50, 20)])[0] == 0                      # same query position: skipped
    assert check([(i, 1000 - i, 15) for i in range(300)])[0] == 0            # anti-diagonal: every candidate is 'after' on t
    assert check([(i * 3, i * 3, 31) for i in range(1000)])[0] == 0          # a perfect diagonal of overlapping anchors
    assert check([(rng.randrange(0, 50), rng.randrange(0, 50), 11) for _ in range(200)].__class__(
        sorted((rng.randrange(0, 50), rng.randrange(0, 50), 11) for _ in range(200))))[0] == 0  # everything inside one band
