// Host build of the product's scratch allocator (ScratchArena / DBuf, lexicmap_amd/csrc/lm_internal.h) over a FAKE device:
// hipMalloc / hipFree / ... are replaced by a bounded host allocator, so the allocator's logic - best fit inside slabs,
// coalescing of free neighbours, slab reuse across the halves of a search, trimming, the out-of-memory answer - runs and is
// checked on the CPU.  Test infrastructure only.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <map>

namespace fake {
static size_t used = 0, limit = 0;
static long mallocs = 0, frees = 0;
static std::map<void *, size_t> live;
static hipError_t Malloc(void **p, size_t n) {
    if (used + n > limit) {
        *p = nullptr;
        return hipErrorOutOfMemory;
    }
    static uintptr_t next_addr = (uintptr_t)1 << 40; // fake device addresses (never dereferenced), 2-MB aligned
    *p = (void *)next_addr;
    next_addr += ((n + ((size_t)2 << 20)) >> 21 << 21) + ((size_t)2 << 20);
    live[*p] = n;
    used += n;
    mallocs++;
    return hipSuccess;
}
static hipError_t Free(void *p) {
    auto it = live.find(p);
    if (it == live.end()) return hipErrorInvalidValue;
    used -= it->second;
    live.erase(it);
    frees++;
    return hipSuccess;
}
static hipError_t MemGetInfo(size_t *fr, size_t *tot) {
    *fr = limit - used;
    *tot = limit;
    return hipSuccess;
}
static hipError_t Ok() { return hipSuccess; }
static hipError_t MemsetAsync(void *p, int v, size_t n, hipStream_t) {
    std::memset(p, v, n);
    return hipSuccess;
}
static const char *ErrStr(hipError_t e) { return e == hipSuccess ? "ok" : "fake out of memory"; }
} // namespace fake

#define hipMalloc(p, n) fake::Malloc((void **)(p), (n))
#define hipFree(p) fake::Free((void *)(p))
#define hipMemGetInfo(a, b) fake::MemGetInfo((a), (b))
#define hipGetLastError() fake::Ok()
#define hipDeviceSynchronize() fake::Ok()
#define hipStreamSynchronize(s) fake::Ok()
#define hipMemsetAsync(p, v, n, s) fake::MemsetAsync((p), (v), (n), (s))
#define hipGetErrorString(e) fake::ErrStr(e)
#define hipHostMalloc(p, n, f) fake::Malloc((void **)(p), (n))
#define hipHostFree(p) fake::Free((void *)(p))

#include "../lexicmap_amd/csrc/lm_internal.h"

extern "C" {
void ah_reset(size_t limit) {
    fake::limit = limit;
}
size_t ah_device_used() { return fake::used; }
long ah_device_mallocs() { return fake::mallocs; }
long ah_device_frees() { return fake::frees; }

void *ah_arena_new() { return new lm::ScratchArena(); }
void ah_arena_delete(void *a) { delete (lm::ScratchArena *)a; }
// returns the block or null on DeviceOOM
void *ah_arena_alloc(void *a, size_t bytes) {
    try {
        return ((lm::ScratchArena *)a)->alloc(bytes);
    } catch (const lm::DeviceOOM &) {
        return nullptr;
    }
}
int ah_arena_release(void *a, void *p) { return ((lm::ScratchArena *)a)->release(p) ? 1 : 0; }
void ah_arena_trim(void *a) { ((lm::ScratchArena *)a)->trim(); }
// the handle's two lane slabs (lm::LaneSlabs) lent to two arenas
void *ah_slabs_new() { return new lm::LaneSlabs(); }
void ah_slabs_delete(void *s) { delete (lm::LaneSlabs *)s; }
int ah_slabs_reserve(void *s, size_t bytes) { return ((lm::LaneSlabs *)s)->reserve(bytes) ? 1 : 0; }
int ah_slabs_assign(void *s, void *a0, void *a1, int lanes) { return ((lm::LaneSlabs *)s)->assign(*(lm::ScratchArena *)a0, *(lm::ScratchArena *)a1, lanes) ? 0 : 1; }
int ah_slabs_unassign(void *s, void *a0, void *a1) { return ((lm::LaneSlabs *)s)->unassign(*(lm::ScratchArena *)a0, *(lm::ScratchArena *)a1) ? 0 : 1; }
long long ah_arena_slab_bytes(void *a) { return ((lm::ScratchArena *)a)->slab_bytes; }
long long ah_arena_live_bytes(void *a) { return ((lm::ScratchArena *)a)->live_bytes; }
long long ah_arena_slab_allocs(void *a) { return ((lm::ScratchArena *)a)->slab_allocs; }
int ah_arena_free_blocks(void *a) { // free-list entries over all slabs (1 per slab when everything is back and coalesced)
    int n = 0;
    for (auto &s : ((lm::ScratchArena *)a)->slabs)
        if (s.base) n += (int)s.free.size();
    return n;
}
int ah_arena_slabs(void *a) {
    int n = 0;
    for (auto &s : ((lm::ScratchArena *)a)->slabs)
        if (s.base) n++;
    return n;
}

// a DBuf<uint8_t> of a search: phase buffers live in the thread's arena, others are plain device allocations
void *ah_dbuf_new(int phase) {
    auto *b = new lm::DBuf<uint8_t>();
    b->phase = phase != 0;
    return b;
}
void ah_dbuf_delete(void *b) { delete (lm::DBuf<uint8_t> *)b; }
// 0 ok, 1 DeviceOOM
int ah_dbuf_ensure(void *b, size_t n, void *arena) {
    lm::tls_arena = (lm::ScratchArena *)arena;
    int rc = 0;
    try {
        ((lm::DBuf<uint8_t> *)b)->ensure(n);
    } catch (const lm::DeviceOOM &) {
        rc = 1;
    }
    lm::tls_arena = nullptr;
    return rc;
}
void ah_dbuf_release(void *b) { ((lm::DBuf<uint8_t> *)b)->release(); }
void *ah_dbuf_ptr(void *b) { return ((lm::DBuf<uint8_t> *)b)->p; }
size_t ah_dbuf_cap(void *b) { return ((lm::DBuf<uint8_t> *)b)->cap; }
int ah_dbuf_in_arena(void *b) { return ((lm::DBuf<uint8_t> *)b)->arena != nullptr; }
long long ah_dbuf_bytes_total() { return lm::g_dbuf_bytes.load(); }
}
