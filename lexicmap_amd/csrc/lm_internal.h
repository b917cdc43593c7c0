// lm_internal.h — shared host-side internals of liblexicmap_hip (handle layout, device buffers, HIP error handling)
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <deque>
#include <mutex>
#include <shared_mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/lexicmap_hip.h"
#include "lm_format.h"
#include "lm_kernels.h"

namespace lm {

struct HipError : std::runtime_error {
    explicit HipError(const std::string &m) : std::runtime_error(m) {}
};
#define HIPCHK(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess)                                                                                \
            throw HipError(std::string(#expr) + ": " + hipGetErrorString(_e) + " at " + __FILE__ + ":" +     \
                           std::to_string(__LINE__));                                                        \
    } while (0)

// a scratch allocation failed: the search path answers by dropping its scratch and halving the query batch part
struct DeviceOOM : HipError {
    explicit DeviceOOM(const std::string &m) : HipError(m) {}
};
inline std::atomic<int64_t> g_dbuf_bytes{0}; // device bytes held by all DBufs of the process (index image + scratch)

// hipMalloc that leaves the device a reserve.  The runtime allocates device memory of its own while a search runs (kernel
// arguments, scratch, signals); with the index, the lane slabs and the overflow slabs a C3 handle held 274 of 288 GB, and one run
// in round 6 ended in "HSA_STATUS_ERROR_OUT_OF_RESOURCES ... Queue aborting" during its warm-up steps - not an error the library
// can catch.  A large request that would leave less than 3 GB free is refused here instead (the callers trim their empty slabs,
// retry, and then halve the batch part: DeviceOOM), on production-size devices only.  (A reserve of 6 GB refused so many overflow
// slabs in one of two C3 runs that the halved parts - they stay halved - made 29 lookup launches per step instead of 17: 8.7
// instead of 8.2 s.  With 3 GB two runs ended at 16 and 18 parts, profiles/r06_c3_reserve3_mem.log.)
static inline hipError_t lm_guarded_malloc(void **p, size_t n) {
    if (n >= ((size_t)16 << 20)) {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess && tot >= ((size_t)64 << 30) && fr < n + ((size_t)3 << 30)) return hipErrorOutOfMemory;
    }
    return hipMalloc(p, n);
}


// Scratch of one index handle that comes and goes with the halves of a search (seeding / alignment): carved
// out of a few large device allocations ("slabs") that stay with the handle, because hipMalloc / hipFree of tens of GB per
// batch part cost seconds (the driver clears the pages).  First fit by address inside a slab, free neighbours coalesce; at
// the end of a half every block is back, so the next half finds whole slabs.  A request no slab can take gets a slab of
// its own; when the device refuses one, the empty slabs are handed back first.
struct ScratchArena {
    struct Slab {
        char *base = nullptr;
        size_t size = 0;
        bool fixed = false;            // one of the handle's lane slabs (LaneSlabs): lent to this arena, never freed by trim()
        std::map<size_t, size_t> free; // offset -> length
    };
    std::mutex mu;
    std::vector<Slab> slabs;
    std::unordered_map<void *, std::pair<int, size_t>> live; // block -> (slab, length)
    int64_t slab_bytes = 0, live_bytes = 0, slab_allocs = 0; // slab_*: what this arena took from the device itself (overflow slabs)
    static constexpr size_t ALIGN = 4096;
    ~ScratchArena() { trim(); }
    // a fixed slab of the handle, whole and free (LaneSlabs::assign)
    void adopt(char *base, size_t size) {
        std::lock_guard<std::mutex> l(mu);
        Slab sl;
        sl.base = base;
        sl.size = size;
        sl.fixed = true;
        sl.free[0] = size;
        put(std::move(sl));
    }
    // the fixed slabs go back to the handle; false (nothing changes) while a block of one of them is live
    bool drop_fixed() {
        std::lock_guard<std::mutex> l(mu);
        for (auto &sl : slabs)
            if (sl.base && sl.fixed && !(sl.free.size() == 1 && sl.free.begin()->second == sl.size)) return false;
        for (auto &sl : slabs)
            if (sl.base && sl.fixed) sl = Slab();
        return true;
    }
    void put(Slab &&sl) {
        for (auto &x : slabs)
            if (!x.base) {
                x = std::move(sl);
                return;
            }
        slabs.push_back(std::move(sl));
    }
    // A slab taken from the device is sized on a coarse geometric grid (2^k x {1, 1.25, 1.5, 1.75}): the pools of a search are re-cut at every
    // batch part with sizes that follow the data (a pool per resident wavefront x the longest problem's expected score), and a
    // request a few per cent above every block freed so far would otherwise get a NEW slab from the device - hipMalloc of
    // several GB clears pages for seconds (measured: 1-3 s steps when the ten WFA chains of a round each re-sized their pools)
    static size_t grid(size_t b) {
        size_t p = ALIGN;
        while (p * 2 <= b) p *= 2;
        const size_t q = p / 4;
        return (b + q - 1) / q * q;
    }
    void *alloc(size_t bytes) { // throws DeviceOOM
        bytes = (bytes + ALIGN - 1) / ALIGN * ALIGN; // (blocks are carved exactly; only a slab of its own is sized on the grid)
        const size_t own = grid(bytes);
        std::lock_guard<std::mutex> l(mu);
        for (int pass = 0; pass < 2; pass++) {
            int bs = -1;
            size_t boff = 0, blen = ~(size_t)0;
            for (size_t si = 0; si < slabs.size(); si++)
                for (auto &f : slabs[si].free)
                    if (f.second >= bytes && f.second < blen) { // best fit over all slabs
                        bs = (int)si;
                        boff = f.first;
                        blen = f.second;
                    }
            if (bs >= 0) {
                Slab &sl = slabs[bs];
                sl.free.erase(boff);
                if (blen > bytes) sl.free[boff + bytes] = blen - bytes;
                void *p = sl.base + boff;
                live[p] = {bs, bytes};
                live_bytes += (int64_t)bytes;
                return p;
            }
            if (pass == 1) break;
            char *base = nullptr;
            size_t got = own;
            hipError_t e = lm_guarded_malloc((void **)&base, got);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                trim_locked();
                e = lm_guarded_malloc((void **)&base, got);
                if (e != hipSuccess && got > bytes) { // (the head-room of the grid is a convenience, not a need)
                    (void)hipGetLastError();
                    got = bytes;
                    e = lm_guarded_malloc((void **)&base, got);
                }
            }
            if (e != hipSuccess) {
                (void)hipGetLastError();
                size_t fr = 0, tot = 0;
                (void)hipMemGetInfo(&fr, &tot);
                throw DeviceOOM("device scratch allocation of " + std::to_string(bytes >> 20) + " MB failed (" +
                                hipGetErrorString(e) + "; " + std::to_string(fr >> 20) + " MB free, " +
                                std::to_string(slab_bytes >> 20) + " MB in scratch slabs, " +
                                std::to_string(g_dbuf_bytes.load() >> 20) + " MB held by this library): use a smaller query batch");
            }
            Slab sl;
            sl.base = base;
            sl.size = got;
            sl.free[0] = got;
            put(std::move(sl));
            slab_bytes += (int64_t)got;
            slab_allocs++;
        }
        throw DeviceOOM("scratch arena: internal error");
    }
    bool release(void *p) {
        std::lock_guard<std::mutex> l(mu);
        auto it = live.find(p);
        if (it == live.end()) return false;
        Slab &sl = slabs[it->second.first];
        size_t off = (size_t)((char *)p - sl.base), len = it->second.second;
        live_bytes -= (int64_t)len;
        live.erase(it);
        auto nx = sl.free.lower_bound(off);
        if (nx != sl.free.end() && off + len == nx->first) { // merge with the free block behind
            len += nx->second;
            nx = sl.free.erase(nx);
        }
        if (nx != sl.free.begin()) {
            auto pv = std::prev(nx);
            if (pv->first + pv->second == off) { // and with the one in front
                pv->second += len;
                return true;
            }
        }
        sl.free[off] = len;
        return true;
    }
    void trim_locked() { // hand the slabs without a live block back to the device (the handle's fixed slabs stay)
        for (auto &sl : slabs)
            if (sl.base && !sl.fixed && sl.free.size() == 1 && sl.free.begin()->second == sl.size) {
                (void)hipFree(sl.base);
                slab_bytes -= (int64_t)sl.size;
                sl = Slab();
            }
    }
    void trim() {
        std::lock_guard<std::mutex> l(mu);
        trim_locked();
    }
};
// The handle's two lane slabs: cut ONCE from the scratch budget at the first search and lent to the lane arenas according
// to the number of lanes a search's budget is divided by.  Why: both lanes used to carve their phase buffers out of one arena
// that grew by hipMalloc on demand - a fresh handle needed three C3 steps to settle (17.5, 14.5, then 12.2 s: hipMalloc /
// hipFree synchronise the device, so one lane's allocation waits for the other lane's persistent WFA kernels), parts were
// halved under the transient pressure and stayed halved, and the serialised measurement step re-cut everything.  With two
// lanes every lane's arena works inside its own slab; with one lane the arena of lane 0 works inside both.  A request no
// slab can take gets an overflow slab from the device (counted: ScratchArena::slab_allocs) that trim() hands back.
struct LaneSlabs {
    char *base[2] = {nullptr, nullptr};
    size_t size[2] = {0, 0};
    int assigned_lanes = 0; // 0: with nobody
    bool asked = false;     // the device was asked once (again after drop())
    // two slabs of bytes / 2 each; false (and holds nothing) when the device refuses: the arenas then work from slabs of
    // their own, i.e. as before
    bool reserve(size_t bytes) {
        drop();
        asked = true;
        const size_t half = bytes / 2 / ScratchArena::ALIGN * ScratchArena::ALIGN;
        if (half == 0) return false;
        for (int i = 0; i < 2; i++) {
            if (hipMalloc((void **)&base[i], half) != hipSuccess) {
                (void)hipGetLastError();
                base[i] = nullptr;
                drop();
                asked = true;
                return false;
            }
            size[i] = half;
        }
        return true;
    }
    // between searches (no live block): lanes == 2 -> one slab each; lanes == 1 -> both to a0.  false: a block is live
    // somewhere (the assignment stays as it is)
    bool assign(ScratchArena &a0, ScratchArena &a1, int lanes) {
        if (!base[0] || lanes == assigned_lanes) return true;
        if (assigned_lanes && !unassign(a0, a1)) return false;
        a0.adopt(base[0], size[0]);
        (lanes == 2 ? a1 : a0).adopt(base[1], size[1]);
        assigned_lanes = lanes;
        return true;
    }
    bool unassign(ScratchArena &a0, ScratchArena &a1) {
        if (assigned_lanes) {
            if (!a0.drop_fixed()) return false;
            if (!a1.drop_fixed()) { // (a0's are back with the handle already: lend them again, nothing changed hands)
                a0.adopt(base[0], size[0]);
                if (assigned_lanes == 1) a0.adopt(base[1], size[1]);
                return false;
            }
        }
        assigned_lanes = 0;
        return true;
    }
    void drop() { // (after unassign)
        for (int i = 0; i < 2; i++) {
            if (base[i]) (void)hipFree(base[i]);
            base[i] = nullptr;
            size[i] = 0;
        }
        assigned_lanes = 0;
        asked = false;
    }
    int64_t bytes() const { return (int64_t)(size[0] + size[1]); }
    ~LaneSlabs() { drop(); }
};
// the arena (and the stream) of the search running on this thread; null outside lm_search_*: plain device allocations
inline thread_local ScratchArena *tls_arena = nullptr;
inline thread_local hipStream_t tls_stream = nullptr;

template <typename T> struct DBuf {
    T *p = nullptr;
    size_t cap = 0;
    ScratchArena *arena = nullptr; // the block lives in this arena (phase == true buffers of a running search)
    bool phase = false;            // released at the end of each half of a search: may live in the handle's arena
    DBuf() = default;
    DBuf(const DBuf &) = delete;
    DBuf &operator=(const DBuf &) = delete;
    ~DBuf() { release(); }
    size_t bytes() const { return cap * sizeof(T); }
    static constexpr size_t ARENA_MIN = (size_t)32 << 20; // smaller buffers stay plain grow-only allocations
    void ensure(size_t n) {
        if (n <= cap && p) return;
        if (p && arena) { // the block goes back while earlier launches of this thread may still read it
            if (tls_stream) (void)hipStreamSynchronize(tls_stream); else (void)hipDeviceSynchronize();
        }
        release();
        size_t want = std::max<size_t>(n + n / 8, 64);
        if (phase && tls_arena && want * sizeof(T) >= ARENA_MIN) {
            p = (T *)tls_arena->alloc(want * sizeof(T));
            arena = tls_arena;
        } else {
            hipError_t e = lm_guarded_malloc((void **)&p, want * sizeof(T));
            if (e != hipSuccess) {
                p = nullptr;
                (void)hipGetLastError();
                if (tls_arena) { // memory parked in empty slabs
                    tls_arena->trim();
                    e = lm_guarded_malloc((void **)&p, want * sizeof(T));
                }
            }
            if (e != hipSuccess) {
                p = nullptr;
                (void)hipGetLastError();
                size_t fr = 0, tot = 0;
                (void)hipMemGetInfo(&fr, &tot);
                throw DeviceOOM("device scratch allocation of " + std::to_string(want * sizeof(T) >> 20) + " MB failed (" +
                                hipGetErrorString(e) + "; " + std::to_string(fr >> 20) + " MB free, " +
                                std::to_string(g_dbuf_bytes.load() >> 20) + " MB held by this library): use a smaller query batch");
            }
        }
        cap = want;
        g_dbuf_bytes += (int64_t)bytes();
    }
    // exactly n elements (the immutable index arrays: no head-room, optionally zero-filled)
    void alloc_exact(size_t n, bool zero = false, hipStream_t st = nullptr) {
        release();
        size_t want = std::max<size_t>(n, 1);
        {
            const hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
            if (e == hipErrorOutOfMemory) { // (its own type: the loader retries without its kept copies)
                p = nullptr;
                (void)hipGetLastError();
                throw DeviceOOM("device allocation of " + std::to_string(want * sizeof(T) >> 20) + " MB failed: out of memory");
            }
            HIPCHK(e);
        }
        cap = want;
        g_dbuf_bytes += (int64_t)bytes();
        if (zero) HIPCHK(hipMemsetAsync(p, 0, want * sizeof(T), st));
    }
    void release() {
        if (p) {
            if (arena) {
                arena->release(p);
            } else {
                (void)hipFree(p);
            }
            g_dbuf_bytes -= (int64_t)bytes();
        }
        arena = nullptr;
        p = nullptr;
        cap = 0;
    }
};

// grow-only pinned host buffer: the destination of the big per-batch device-to-host copies (pageable targets go through
// a staging copy at a fraction of the PCIe rate)
template <typename T> struct PBuf {
    T *p = nullptr;
    size_t cap = 0;
    PBuf() = default;
    PBuf(const PBuf &) = delete;
    PBuf &operator=(const PBuf &) = delete;
    ~PBuf() {
        if (p) (void)hipHostFree(p);
    }
    void ensure(size_t n) {
        if (n <= cap && p) return;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        size_t want = std::max<size_t>(n + n / 8, 64);
        HIPCHK(hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocDefault));
        cap = want;
    }
};

static inline double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct ProfEntry {
    std::string name;
    int64_t launches = 0;
    double ms = 0;
    int64_t bytes = 0;
};

} // namespace lm

using namespace lm;

extern thread_local std::string g_open_error; // text of the last failed open/build (lm_last_error(NULL))
struct lm_index;
void lm_fill_gap_lut(lm_index *ix);
void lm_set_scratch_budget(lm_index *ix);
void lm_reserve_lane_slabs(lm_index *ix);

namespace lm {
struct Work;
struct AlignCtx;

// Two-pass construction of the packed seed image in HBM (lm_seedpack.hip): every seed is shown twice, in any order and
// in batches of any size, as (mask, k-mer, value in the reference layout): count() sizes the partitions, place() stores.
struct SeedPacker {
    lm_index *ix = nullptr;
    int a = 0, P1 = 0, key_bits = 0, gid_bits = 0, pos_bits = 0;
    int64_t n_main = 0, n_out = 0;
    DBuf<unsigned long long> out_cnt; // [2M+1] outlier counts, then cursors
    bool placing = false;
    // ix->view.masks / pfx_first / batch_first / shard fields and ix->host.{k, M, mask_prefix, anchor_prefix} must be set
    void begin(lm_index *ix, int64_t local_genomes, int64_t max_genome_len);
    void count(const uint16_t *mask, const uint64_t *kmer, const uint64_t *val, int64_t n);
    void end_count();
    void place(const uint16_t *mask, const uint64_t *kmer, const uint64_t *val, int64_t n);
    void finish(); // sorts the partitions, fills ix->view / ix->n_seeds / ix->seed_bytes
};
} // namespace lm

// Experiment / A-B switches of the kernels, read from the environment ONCE when the handle is created (a search never calls
// getenv for them). Defaults are the measured winners; the alternatives stay for the profiles that justify them.
#define LM_WFA_CLASSES 5 /* length classes of the WFA problems of a round (run_wfa) */
struct lm_tune {
    // WFA per length class (<= 2 kb, <= 8 kb, <= 32 kb, <= 65 kb, longer): the ring width (cells per lane x 64 diagonals) a
    // problem starts with, and whether the kernel reads the sequences through sliding LDS windows (1) or keeps them whole in
    // LDS (0; impossible beyond 65 kb).  LM_WFA_FIRST_NC="2,2,4,8,8", LM_WFA_WIN="00101" override.
    int wfa_first_nc[LM_WFA_CLASSES] = {2, 2, 4, 8, 8};
    int wfa_win[LM_WFA_CLASSES] = {0, 0, 1, 0, 1};
    int chain1_wave = 1;     // seed chaining of pairs with many anchors by a wavefront each
    int pa_filter_roll = 1;  // k_pa_filter: a lane takes consecutive window positions (immediate funnel shifts over its own 64-base string); LM_PA_FILTER_ROLL=0: every 64th position, words passed between lanes
    int pa_seg_by_group = 1; // candidate segments by task group + XCD-local k_pa_search
    int wfa_resident_pct = 100; // share of the CUs' wavefront slots / LDS the persistent WFA kernels take (75 / 50 % measured in round 3: no gain / -20 %)
    int arena_reserve_pct = 90; // LM_ARENA_RESERVE_PCT: share of the scratch budget cut into the two lane slabs when a production-size index is opened, else at the first search (LaneSlabs; 0: slabs on demand as in round 4)
    int two_lanes = 1;       // two parts of a batch searched side by side, each with half of the scratch budget (LM_TWO_LANES=0: one after the other)
    int lookup_flat = 1;     // anchors emitted with the lanes over the output (k_lookup_emit_flat); LM_LOOKUP_FLAT=0: one lane per lookup
    int wfa_r16 = 1;         // 16-bit ring cells in the whole-sequence WFA kernels of 128 / 256 diagonals (LM_WFA_R16=0: 32-bit)
    int wfa_serial = 0;      // the WFA length classes one after the other (lm_profile_exclusive: exclusive kernel timings)
    int no_pipeline = 0;     // no pseudo-alignment producer beside extend / WFA (lm_profile_exclusive)
    lm_tune() {
        if (const char *e = getenv("LM_WFA_FIRST_NC")) {
            int v[LM_WFA_CLASSES];
            if (sscanf(e, "%d,%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3], &v[4]) == LM_WFA_CLASSES)
                for (int c = 0; c < LM_WFA_CLASSES; c++)
                    if (v[c] == 1 || v[c] == 2 || v[c] == 4 || v[c] == 8 || v[c] == 16) wfa_first_nc[c] = v[c];
        }
        if (const char *e = getenv("LM_WFA_WIN"))
            for (int c = 0; c < LM_WFA_CLASSES && e[c]; c++) wfa_win[c] = e[c] == '1';
        if (const char *e = getenv("LM_WFA_R16")) wfa_r16 = atoi(e) != 0;
        if (const char *e = getenv("LM_LOOKUP_FLAT")) lookup_flat = atoi(e) != 0;
        if (const char *e = getenv("LM_TWO_LANES")) two_lanes = atoi(e) != 0;
        if (const char *e = getenv("LM_ARENA_RESERVE_PCT")) arena_reserve_pct = std::max(0, std::min(100, atoi(e)));
        if (const char *e = getenv("LM_PA_FILTER_ROLL")) pa_filter_roll = atoi(e) != 0;
    }
};

struct lm_index {
    lm_tune tune;
    lm::Work *work = nullptr;       // device scratch reused across calls (grow-only)
    lm::AlignCtx *actx[3] = {nullptr, nullptr, nullptr}; // consumer (glue, extend, first WFA passes), pseudo-alignment producer, WFA tail
    // second lane: two parts of a large batch are searched side by side (the seeding / anchor kernels of one beside the
    // WFA launches of the other), each with half of the scratch budget, its own scratch, streams and rocPRIM storage
    lm::Work *work1 = nullptr;
    lm::AlignCtx *actx1[3] = {nullptr, nullptr, nullptr};
    hipStream_t st_b = nullptr, st2_b = nullptr;
    int active_lanes = 1;
    int budget_lanes = 1; // what BUDGET() divides the scratch budget by (= active_lanes, but 2 in the serialised measurement step of a two-lane search)
    std::mutex mu;                  // one in-flight call per handle
    HostIndex host;
    lm_options opt;
    int device = 0;
    hipStream_t st = nullptr, st2 = nullptr; // st2: the pseudo-alignment producer of the alignment pipeline
    std::string err;
    // HBM image
    DBuf<uint64_t> d_masks, d_pk_keys, d_pk_vals, d_out_kmers, d_out_vals, d_g_bg;
    DBuf<uint32_t> d_part_tab, d_g_keep;
    DBuf<int32_t> d_pfx_first, d_g_len, d_g2local;
    DBuf<int64_t> d_md_off, d_out_off, d_g_off, d_batch_first;
    int64_t n_seeds = 0, n_seeds_outlier = 0, seed_bytes = 0; // resident seeds; bytes of the whole seed image
    int64_t scratch_budget = 0; // device memory left for per-batch scratch once the index image is resident
    DBuf<uint8_t> d_gbits;
    DBuf<float> d_gap_lut;
    int gap_lut_n = 0;
    DevIndexView view;
    int64_t hbm_bytes = 0;
    // scratch
    LaneSlabs lane_slabs;    // the two fixed slabs the lane arenas work in (LM_ARENA_RESERVE_PCT of the scratch budget)
    ScratchArena arena[2];   // phase buffers of the searches on this handle, one arena per lane (destroyed after work / actx)
    DBuf<uint8_t> tmp, tmp2, tmp_b, tmp2_b; // rocPRIM temporary storage (per stream)
    // profiling
    bool prof = false;
    std::mutex prof_mu;
    std::vector<ProfEntry> prof_entries;
    std::vector<lm_kernel_time> prof_out;
    struct Pending {
        int entry;
        hipEvent_t a, b;
    };
    std::vector<Pending> pending;
    // genome lookup
    std::unordered_map<uint64_t, int> bg2local;
    // names of a synthetic set (a function of the genome number), made on first use and kept with the handle: ONE copy per genome
    // (lm_merge_sharded made two strings per ROW: ten million allocations per C3 step on the merging rank)
    std::shared_mutex syn_mu;
    std::unordered_map<uint64_t, std::pair<const char *, const char *>> syn_names; // (genome numbers beyond the set: bench.py's emulated shards)
    std::unique_ptr<std::atomic<const char *>[]> syn_dense;                         // [2 * synth_genomes]: id, sequence id; NULL until first use
    std::deque<std::string> syn_store;
};

