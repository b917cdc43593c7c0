"""lexicmap_amd/csrc/lm_pa_chain_bt_core.h + lm_pa_clear_tile.h (product headers): the backtrack of Chainer2 - regions left and
right of every chain, the walk from anchor to predecessor, the chain statistics - by a wavefront (region scans by 64 lanes,
the walk out of 64-anchor tiles in LDS) instead of one lane chasing pointers through global memory; on the host SIMT emulator
against lm_run_chain2 (lm_algos.h): every chain, every field, the order after the sort by QBegin."""
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
        path = os.path.join(EMU, "libpa_chain_bt_emu.so")
        srcs = [os.path.join(EMU, "pa_chain_bt_emu.cpp"), os.path.join(root, "lexicmap_amd", "csrc", "lm_pa_chain_bt_core.h"), os.path.join(root, "lexicmap_amd", "csrc", "lm_pa_clear_tile.h"),
                os.path.join(EMU, "simt_emu.h"), os.path.join(root, "lexicmap_amd", "csrc", "lm_algos.h")]
        if not os.path.exists(path) or os.path.getmtime(path) < max(os.path.getmtime(s) for s in srcs):
            subprocess.check_call(["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-o", path, srcs[0]])
        _lib = C.CDLL(path)
    return _lib


def check(anchors, max_gap=20, band_base=100, band_count=50, min_score=30, min_align_len=20, pident=15.0):
    """-> (differences, chains of the reference)"""
    n = len(anchors)
    qb = (C.c_int32 * n)(*[a[0] for a in anchors])
    tb = (C.c_int32 * n)(*[a[1] for a in anchors])
    ln = (C.c_uint8 * n)(*[a[2] for a in anchors])
    nch = C.c_int()
    bad = lib().pcb_emu_check(qb, tb, ln, n, max_gap, band_base, band_count, min_score, min_align_len, C.c_double(pident), C.byref(nch))
    return bad, nch.value


def broken(rng, n, pieces):
    """`pieces` colinear runs far apart in the target: several chains, regions between and around them"""
    out = []
    per = max(2, n // pieces)
    q = 0
    for p in range(pieces):
        run = colinear(rng, per)
        dq = q - run[0][0]
        dt = rng.randrange(0, 200000)
        out += [(a[0] + dq, a[1] + dt, a[2]) for a in run]
        q = out[-1][0] + rng.randint(20, 400)
    return sorted(out)


@pytest.mark.parametrize("n,seed", [(2, 1), (3, 2), (9, 3), (64, 4), (65, 5), (130, 6), (700, 7), (3000, 8)])
def test_wavefront_backtrack_equals_lm_run_chain2(n, seed):
    rng = random.Random(seed)
    total = 0
    for rep in range(3):
        bad, nch = check(colinear(rng, n + rep))
        assert bad == 0
        total += nch
    if n >= 64:
        assert total >= 1


@pytest.mark.parametrize("n,pieces,seed", [(300, 3, 11), (1200, 7, 12), (2500, 20, 13), (800, 40, 14)])
def test_several_chains_and_the_regions_between_them(n, pieces, seed):
    rng = random.Random(seed)
    bad, nch = check(broken(rng, n, pieces))
    assert bad == 0 and nch >= 2
    # stricter filters drop chains on the way (min_align_len / pident breaks inside the walk)
    assert check(broken(rng, n, pieces), min_align_len=200)[0] == 0
    assert check(broken(rng, n, pieces), pident=80.0, min_score=60)[0] == 0


def test_degenerate_inputs():
    rng = random.Random(21)
    assert check([(10, 10, 20), (10, 50, 20)])[0] == 0
    assert check([(i, 1000 - i, 15) for i in range(300)])[0] == 0            # anti-diagonal: nothing chains
    assert check([(i * 3, i * 3, 31) for i in range(1000)])[0] == 0          # one perfect diagonal of overlapping anchors
    assert check(sorted((rng.randrange(0, 50), rng.randrange(0, 50), 11) for _ in range(200)))[0] == 0
    assert check(colinear(rng, 500), min_score=1 << 30) == (0, 0)            # nothing reaches the score: no chain
    assert check(colinear(rng, 500), min_score=1)[0] == 0                    # every region is walked down to single anchors


def check_clear(anchors, K=31):
    n = len(anchors)
    qb = (C.c_int32 * n)(*[a[0] for a in anchors])
    tb = (C.c_int32 * n)(*[a[1] for a in anchors])
    ln = (C.c_uint8 * n)(*[a[2] for a in anchors])
    kept = C.c_int()
    return lib().pcc_emu_check(qb, tb, ln, n, K, C.byref(kept)), kept.value


def dense(rng, n, per_pos=(1, 3), step=(1, 4)):
    """pseudo-alignment-like anchors: along a diagonal, a few per query position, lengths 11-31 - most are nested in an earlier one"""
    out, q = [], 0
    while len(out) < n:
        for _ in range(rng.randint(*per_pos)):
            out.append((q, q + rng.choice((0, 0, 0, 1, -1, 500)), rng.randint(11, 31)))
        q += rng.randint(*step)
    return sorted(out[:n])


@pytest.mark.parametrize("n,seed", [(1, 1), (2, 2), (63, 3), (64, 4), (65, 5), (129, 6), (1000, 7), (4000, 8)])
def test_clear_marks_from_lds_tiles_equal_lm_clear_sorted(n, seed):
    rng = random.Random(seed)
    bad, kept = check_clear(dense(rng, n))
    assert bad == 0
    if n >= 64:
        assert kept < n  # (something is nested)
    assert check_clear(colinear(rng, n))[0] == 0


def test_clear_marks_when_the_candidates_reach_beyond_the_halo():
    """more than 64 anchors within the K - len bases before an anchor: the scan goes on in global memory"""
    rng = random.Random(31)
    crowded = sorted([(100 + rng.randint(0, 3), 100 + rng.randint(0, 900), 11) for _ in range(300)] + [(104, 104, 31), (104, 600, 12)])
    assert check_clear(crowded)[0] == 0
    assert check_clear(dense(rng, 2000, per_pos=(20, 40), step=(1, 2)))[0] == 0
    assert check_clear(dense(rng, 500), K=15)[0] == 0
    assert check_clear([(5, 5, 31)] * 200)[0] == 0  # identical anchors: every one but the first is nested


def test_a_tile_serves_tens_of_walk_steps():
    """the point of the LDS tile: the walk along a chain of 3000 anchors reloads it once per ~40 steps (the lane-0 backtrack
    it replaces makes two dependent global loads per step)"""
    rng = random.Random(41)
    bad, nch = check(colinear(rng, 3000))
    assert bad == 0 and nch >= 1
    steps, tiles = C.c_long(), C.c_long()
    lib().pcb_emu_counts(C.byref(steps), C.byref(tiles))
    assert steps.value >= 1000 and tiles.value * 20 <= steps.value, (steps.value, tiles.value)
