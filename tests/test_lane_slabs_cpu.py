"""lm::LaneSlabs + lm::ScratchArena (lexicmap_amd/csrc/lm_internal.h, product code): the scratch of a two-lane handle as two
fixed slabs cut once from the budget.  Over a fake device: two lanes carving jittered, C3-like pools concurrently never reach
the device after the reservation (round 4's shared growing arena took three C3 steps of hipMalloc stalls to settle and halved
batch parts under the transient pressure); a single-lane search on the same handle gets both slabs; what no slab can take goes
to an overflow slab that trim() hands back; the slabs cannot change hands while a block is live."""
import ctypes as C
import os
import random
import subprocess
import threading

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libarena_host.so")
SRC = os.path.join(HERE, "arena_host.cpp")
HDR = os.path.join(os.path.dirname(HERE), "lexicmap_amd", "csrc", "lm_internal.h")
MB = 1 << 20
GB = 1 << 30


class _Names:
    """the la_* names of this file on the ah_* exports of tests/arena_host.cpp"""
    MAP = {"la_arena_new": "ah_arena_new", "la_arena_delete": "ah_arena_delete", "la_alloc": "ah_arena_alloc", "la_release": "ah_arena_release",
           "la_trim": "ah_arena_trim", "la_live_bytes": "ah_arena_live_bytes", "la_overflow_allocs": "ah_arena_slab_allocs",
           "la_slabs_new": "ah_slabs_new", "la_slabs_delete": "ah_slabs_delete", "la_slabs_reserve": "ah_slabs_reserve",
           "la_slabs_assign": "ah_slabs_assign", "la_slabs_unassign": "ah_slabs_unassign", "la_reset": "ah_reset",
           "la_device_used": "ah_device_used", "la_device_mallocs": "ah_device_mallocs"}

    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, k):
        return getattr(self._lib, self.MAP[k])


@pytest.fixture(scope="module")
def L():
    if not os.path.isdir("/opt/rocm/include"):
        pytest.skip("HIP headers not installed")
    srcs = [SRC, HDR]
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                               "-o", LIB, SRC])
    lib = _Names(C.CDLL(LIB))
    for f in ("la_arena_new", "la_alloc", "la_slabs_new"):
        getattr(lib, f).restype = C.c_void_p
    lib.la_alloc.argtypes = [C.c_void_p, C.c_size_t]
    lib.la_release.argtypes = [C.c_void_p, C.c_void_p]
    for f in ("la_arena_delete", "la_trim", "la_slabs_delete"):
        getattr(lib, f).argtypes = [C.c_void_p]
    for f in ("la_live_bytes", "la_overflow_allocs"):
        getattr(lib, f).restype = C.c_longlong
        getattr(lib, f).argtypes = [C.c_void_p]
    lib.la_slabs_reserve.argtypes = [C.c_void_p, C.c_size_t]
    lib.la_slabs_assign.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.la_slabs_unassign.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.la_release.restype = C.c_int
    lib.la_reset.argtypes = [C.c_size_t]
    lib.la_device_used.restype = C.c_size_t
    lib.la_device_mallocs.restype = C.c_long
    return lib


# the phase buffers of one lane's search part at C3 as shares of the lane's budget: the seeding half, then
# the alignment half (two pseudo-alignment chunks in flight, window buffers, WFA pools of the ten chains, the fallback)
SEEDING = [0.02] * 6 + [0.03, 0.03, 0.05, 0.05]
ALIGN = [0.11, 0.11, 0.025, 0.025, 0.07, 0.05, 0.04, 0.03, 0.02, 0.02, 0.01, 0.01, 0.08]


def run_part(L, arena, budget, rng, scale):
    for half in (SEEDING, ALIGN):
        ps = []
        order = list(half)
        rng.shuffle(order)          # the worker threads of a half allocate in whatever order they get there
        for share in order:
            p = L.la_alloc(arena, int(share * scale * rng.uniform(0.9, 1.0) * budget))
            assert p is not None
            ps.append(p)
        rng.shuffle(ps)
        for p in ps:
            assert L.la_release(arena, p) == 1


def test_two_lanes_never_reach_the_device_after_the_reservation_and_one_lane_gets_both_slabs(L):
    L.la_reset(160 * GB)
    slabs, a0, a1 = L.la_slabs_new(), L.la_arena_new(), L.la_arena_new()
    budget = 120 * GB
    assert L.la_slabs_reserve(slabs, budget) == 1
    m0 = L.la_device_mallocs()
    assert m0 == 2 and L.la_device_used() == budget
    assert L.la_slabs_assign(slabs, a0, a1, 2) == 0

    def lane(arena, seed):
        rng = random.Random(seed)
        for _ in range(12):     # parts of different sizes, taken in whatever order the lanes get to them
            run_part(L, arena, budget // 2, rng, rng.uniform(0.6, 1.0))
    ts = [threading.Thread(target=lane, args=(a, s)) for a, s in ((a0, 1), (a1, 2))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert L.la_device_mallocs() == m0 and L.la_overflow_allocs(a0) == 0 and L.la_overflow_allocs(a1) == 0
    assert L.la_live_bytes(a0) == 0 and L.la_live_bytes(a1) == 0
    # the serialised measurement step / a single-part batch: one lane, the whole budget, both slabs - still nothing new
    assert L.la_slabs_assign(slabs, a0, a1, 1) == 0
    rng = random.Random(3)
    for _ in range(4):
        run_part(L, a0, budget, rng, rng.uniform(0.7, 1.0))
    assert L.la_device_mallocs() == m0 and L.la_overflow_allocs(a0) == 0
    # ... and back to two lanes
    assert L.la_slabs_assign(slabs, a0, a1, 2) == 0
    run_part(L, a1, budget // 2, rng, 1.0)
    assert L.la_device_mallocs() == m0
    assert L.la_slabs_unassign(slabs, a0, a1) == 0
    L.la_arena_delete(a0)
    L.la_arena_delete(a1)
    L.la_slabs_delete(slabs)
    assert L.la_device_used() == 0


def test_overflow_slabs_and_ownership_rules(L):
    L.la_reset(10 * GB)
    slabs, a0, a1 = L.la_slabs_new(), L.la_arena_new(), L.la_arena_new()
    assert L.la_slabs_reserve(slabs, 4 * GB) == 1 and L.la_slabs_assign(slabs, a0, a1, 2) == 0
    m0 = L.la_device_mallocs()
    big = L.la_alloc(a0, 3 * GB)                         # larger than lane 0's 2-GB slab: an overflow slab of its own
    assert big is not None and L.la_overflow_allocs(a0) == 1 and L.la_device_mallocs() == m0 + 1
    small = L.la_alloc(a0, 100 * MB)                     # ... while ordinary requests stay inside the slab
    assert small is not None and L.la_device_mallocs() == m0 + 1
    assert L.la_slabs_assign(slabs, a0, a1, 1) == 1      # the slabs cannot change hands while a block is live
    assert L.la_release(a0, small) == 1 and L.la_release(a0, big) == 1
    L.la_trim(a0)                                        # the empty overflow slab goes back, the handle's slabs stay
    assert L.la_device_used() == 4 * GB
    assert L.la_slabs_assign(slabs, a0, a1, 1) == 0
    whole = [L.la_alloc(a0, 2 * GB - 4096), L.la_alloc(a0, 2 * GB - 4096)]   # one lane: both slabs
    assert all(whole) and L.la_device_mallocs() == m0 + 1
    assert L.la_alloc(a0, 7 * GB) is None                # neither a slab nor the device (10 GB, 4 held) has room: DeviceOOM
    for p in whole:
        L.la_release(a0, p)
    # a device that cannot hold the reservation: nothing is held, the arenas then work from overflow slabs (= round 4's behaviour)
    L.la_slabs_unassign(slabs, a0, a1)
    assert L.la_slabs_reserve(slabs, 64 * GB) == 0 and L.la_device_used() == 0
    assert L.la_slabs_assign(slabs, a0, a1, 2) == 0
    p = L.la_alloc(a1, 1 * GB)
    assert p is not None and L.la_overflow_allocs(a1) == 1
    L.la_release(a1, p)
    L.la_arena_delete(a0)
    L.la_arena_delete(a1)
    L.la_slabs_delete(slabs)
    assert L.la_device_used() == 0
