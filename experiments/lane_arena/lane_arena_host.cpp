// experiments/lane_arena/lane_arena_host.cpp - lm_lane_arena.h over the fake device of tests/arena_host.cpp (same macros), for
// tests/test_lane_arena_staged_cpu.py.  Staged code, not the product.
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
    static uintptr_t next_addr = (uintptr_t)1 << 40;
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
static hipError_t Ok() { return hipSuccess; }
} // namespace fake

#define hipMalloc(p, n) fake::Malloc((void **)(p), (n))
#define hipFree(p) fake::Free((void *)(p))
#define hipGetLastError() fake::Ok()

#include <stdexcept>
#include <string>
namespace lm {
struct DeviceOOM : std::runtime_error {
    explicit DeviceOOM(const std::string &m) : std::runtime_error(m) {}
};
} // namespace lm
#include "lm_lane_arena.h"

extern "C" {
void la_reset(size_t limit) { fake::limit = limit; }
size_t la_device_used() { return fake::used; }
long la_device_mallocs() { return fake::mallocs; }
long la_device_frees() { return fake::frees; }
void *la_arena_new() { return new lm::LaneArena(); }
void la_arena_delete(void *a) { delete (lm::LaneArena *)a; }
void *la_alloc(void *a, size_t bytes) {
    try {
        return ((lm::LaneArena *)a)->alloc(bytes);
    } catch (const lm::DeviceOOM &) {
        return nullptr;
    }
}
int la_release(void *a, void *p) { return ((lm::LaneArena *)a)->release(p) ? 1 : 0; }
void la_trim(void *a) { ((lm::LaneArena *)a)->trim(); }
long long la_live_bytes(void *a) { return ((lm::LaneArena *)a)->live_bytes; }
long long la_overflow_allocs(void *a) { return ((lm::LaneArena *)a)->overflow_allocs; }
void *la_slabs_new() { return new lm::LaneSlabs(); }
void la_slabs_delete(void *s) {
    ((lm::LaneSlabs *)s)->drop();
    delete (lm::LaneSlabs *)s;
}
int la_slabs_reserve(void *s, size_t bytes) { return ((lm::LaneSlabs *)s)->reserve(bytes) ? 1 : 0; }
int la_slabs_assign(void *s, void *a0, void *a1, int lanes) {
    try {
        ((lm::LaneSlabs *)s)->assign(*(lm::LaneArena *)a0, *(lm::LaneArena *)a1, lanes);
        return 0;
    } catch (const std::exception &) {
        return 1;
    }
}
void la_slabs_unassign(void *s, void *a0, void *a1) { ((lm::LaneSlabs *)s)->unassign(*(lm::LaneArena *)a0, *(lm::LaneArena *)a1); }
}
