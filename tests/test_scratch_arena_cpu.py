"""The product's scratch allocator (ScratchArena / DBuf, lexicmap_amd/csrc/lm_internal.h) built for the host over a fake
device (tests/arena_host.cpp): slabs are reused across the halves of a search, free neighbours coalesce, empty slabs are
handed back when the device refuses an allocation, and an allocation that cannot be served raises DeviceOOM (which the
search answers by halving the batch part)."""
import ctypes as C
import os
import random
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "arena_host.cpp")
HDR = os.path.join(os.path.dirname(HERE), "lexicmap_amd", "csrc", "lm_internal.h")
LIB = os.path.join(HERE, "libarena_host.so")
MB = 1 << 20


def grid(n):
    """ScratchArena::grid: a slab taken from the device for a request is sized 2^k x {1, 1.25, 1.5, 1.75} (multiples of 4096); the block itself is carved exactly (4096-byte units)"""
    b = (n + 4095) // 4096 * 4096
    p = 4096
    while p * 2 <= b:
        p *= 2
    q = p // 4
    return (b + q - 1) // q * q


@pytest.fixture(scope="module")
def L():
    if not os.path.isdir("/opt/rocm/include"):
        pytest.skip("HIP headers not installed")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(SRC), os.path.getmtime(HDR)):
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__",
                               "-I/opt/rocm/include", "-o", LIB, SRC])
    lib = C.CDLL(LIB)
    for f in ("ah_arena_new", "ah_arena_alloc", "ah_dbuf_new", "ah_dbuf_ptr"):
        getattr(lib, f).restype = C.c_void_p
    lib.ah_arena_alloc.argtypes = [C.c_void_p, C.c_size_t]
    lib.ah_arena_release.argtypes = [C.c_void_p, C.c_void_p]
    for f in ("ah_arena_delete", "ah_arena_trim", "ah_dbuf_delete", "ah_dbuf_release"):
        getattr(lib, f).argtypes = [C.c_void_p]
    for f in ("ah_arena_slab_bytes", "ah_arena_live_bytes", "ah_arena_slab_allocs", "ah_dbuf_bytes_total"):
        getattr(lib, f).restype = C.c_longlong
    for f in ("ah_arena_slab_bytes", "ah_arena_live_bytes", "ah_arena_slab_allocs", "ah_arena_free_blocks", "ah_arena_slabs",
              "ah_dbuf_ptr", "ah_dbuf_cap", "ah_dbuf_in_arena"):
        getattr(lib, f).argtypes = [C.c_void_p]
    lib.ah_dbuf_cap.restype = C.c_size_t
    lib.ah_device_used.restype = C.c_size_t
    lib.ah_device_mallocs.restype = C.c_long
    lib.ah_device_frees.restype = C.c_long
    lib.ah_reset.argtypes = [C.c_size_t]
    lib.ah_dbuf_ensure.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    return lib


def test_blocks_never_overlap_and_everything_coalesces_back(L):
    L.ah_reset(4096 * MB)
    a = L.ah_arena_new()
    rng = random.Random(7)
    live = {}  # ptr -> size
    for step in range(3000):
        if live and (rng.random() < 0.45 or len(live) > 60):
            p = rng.choice(list(live))
            assert L.ah_arena_release(a, p) == 1
            del live[p]
        else:
            n = rng.choice([1, 4096, 33 * MB, 64 * MB, 200 * MB, rng.randrange(1, 300 * MB)])
            p = L.ah_arena_alloc(a, n)
            if p is None:      # the fake device is full: legal, nothing must have changed
                continue
            assert p % 4096 == 0 and p not in live
            live[p] = (n + 4095) // 4096 * 4096
            iv = sorted(live.items())
            for (p0, n0), (p1, _n1) in zip(iv, iv[1:]):
                assert p0 + n0 <= p1, "live blocks overlap"
        assert L.ah_arena_live_bytes(a) == sum(live.values())
        assert L.ah_arena_slab_bytes(a) == L.ah_device_used()
    for p in list(live):
        assert L.ah_arena_release(a, p) == 1
    assert L.ah_arena_release(a, 12345) == 0                    # not a block of this arena
    assert L.ah_arena_live_bytes(a) == 0
    assert L.ah_arena_free_blocks(a) == L.ah_arena_slabs(a)     # one free block per slab: neighbours coalesced
    L.ah_arena_trim(a)
    assert L.ah_arena_slabs(a) == 0 and L.ah_device_used() == 0
    L.ah_arena_delete(a)


def test_the_halves_of_a_search_alternate_without_device_allocations(L):
    """seeding half: many buffers of a few hundred MB; alignment half: a few large ones; every block goes back at the end of
    a half.  After the first part no device allocation happens any more (what cost 24 s of a 50-s step at C3)."""
    L.ah_reset(64 * 1024 * MB)
    a = L.ah_arena_new()
    rng = random.Random(3)

    def half(sizes):
        ps = [L.ah_arena_alloc(a, s) for s in sizes]
        assert all(p is not None for p in ps)
        for p in ps:
            assert L.ah_arena_release(a, p) == 1

    def part(scale):
        half([int(rng.uniform(0.8, 1.0) * scale * 400 * MB) for _ in range(40)])              # seeding
        half([int(scale * 9000 * MB), int(scale * 3000 * MB)] + [int(scale * 700 * MB)] * 6)  # alignment
    part(1.0)
    m0 = L.ah_device_mallocs()
    for _ in range(6):
        part(rng.uniform(0.7, 1.0))   # later parts are never larger than what the slabs have seen
    assert L.ah_device_mallocs() == m0
    assert L.ah_arena_live_bytes(a) == 0
    L.ah_arena_delete(a)
    assert L.ah_device_used() == 0


def test_pools_whose_sizes_follow_the_data_do_not_reach_the_device_again(L):
    """the WFA chains of a round size their pools by (wavefronts x expected score of the longest problem): a few per cent up
    or down from round to round and from part to part.  Without the size grid every request slightly above all freed blocks
    got a new slab (hipMalloc of GBs = seconds); with it the blocks of the first rounds serve the later ones."""
    L.ah_reset(256 * 1024 * MB)
    a = L.ah_arena_new()
    rng = random.Random(11)
    base = [7300, 660, 1700, 5200, 5900, 14000, 2500, 900, 350, 12000]   # MB: the ten chains of a c3-shaped round

    def round_(jit):
        ps = [L.ah_arena_alloc(a, int(b * MB * rng.uniform(1 - jit, 1 + jit))) for b in base]
        assert all(p is not None for p in ps)
        for p in ps:
            assert L.ah_arena_release(a, p) == 1

    for _ in range(4):
        round_(0.08)
    m0 = L.ah_device_mallocs()
    for _ in range(40):
        round_(0.08)
    assert L.ah_device_mallocs() - m0 <= 4      # (a size that crosses a grid step may still want one more slab)
    L.ah_arena_delete(a)
    assert L.ah_device_used() == 0


def test_empty_slabs_are_handed_back_before_giving_up_and_oom_is_reported(L):
    L.ah_reset(1100 * MB)
    a = L.ah_arena_new()
    ps = [L.ah_arena_alloc(a, 100 * MB) for _ in range(8)]       # eight slabs of 100 MB (112 MB on the grid)
    assert all(ps) and L.ah_arena_slabs(a) == 8 and L.ah_device_used() == 8 * grid(100 * MB)
    for p in ps:
        L.ah_arena_release(a, p)
    big = L.ah_arena_alloc(a, 900 * MB)                          # fits only if the empty slabs go back first
    assert big is not None and L.ah_arena_slabs(a) == 1 and L.ah_device_used() == grid(900 * MB) == 1024 * MB
    assert L.ah_arena_alloc(a, 200 * MB) is None                 # DeviceOOM, arena unchanged
    assert L.ah_arena_live_bytes(a) == 900 * MB
    L.ah_arena_release(a, big)
    L.ah_arena_delete(a)
    assert L.ah_device_used() == 0


def test_dbuf_phase_buffers_use_the_arena_and_small_or_plain_ones_do_not(L):
    L.ah_reset(2048 * MB)
    a = L.ah_arena_new()
    base = L.ah_dbuf_bytes_total()
    big, small, plain = L.ah_dbuf_new(1), L.ah_dbuf_new(1), L.ah_dbuf_new(0)
    assert L.ah_dbuf_ensure(big, 100 * MB, a) == 0 and L.ah_dbuf_in_arena(big) == 1
    assert L.ah_dbuf_cap(big) >= 100 * MB                         # 1/8 head room
    assert L.ah_dbuf_ensure(small, 1 * MB, a) == 0 and L.ah_dbuf_in_arena(small) == 0   # below 32 MB: plain
    assert L.ah_dbuf_ensure(plain, 100 * MB, a) == 0 and L.ah_dbuf_in_arena(plain) == 0  # not a phase buffer
    assert L.ah_dbuf_ensure(big, 10 * MB, a) == 0                 # grow-only: no change
    p0 = L.ah_dbuf_ptr(big)
    assert L.ah_dbuf_ensure(big, 300 * MB, a) == 0 and L.ah_dbuf_in_arena(big) == 1
    L.ah_dbuf_release(big)
    assert L.ah_arena_live_bytes(a) == 0
    again = L.ah_dbuf_new(1)
    assert L.ah_dbuf_ensure(again, 300 * MB, a) == 0              # carved from the slab the released buffer left
    assert L.ah_arena_slab_allocs(a) == 2 and p0 is not None
    assert L.ah_dbuf_ensure(again, 4000 * MB, a) == 1             # beyond the fake device: DeviceOOM, buffer empty
    assert L.ah_dbuf_ptr(again) is None and L.ah_dbuf_cap(again) == 0
    for b in (big, small, plain, again):
        L.ah_dbuf_delete(b)
    assert L.ah_dbuf_bytes_total() == base
    L.ah_arena_delete(a)
    assert L.ah_device_used() == 0
