// lm_internal.h — shared host-side internals of liblexicmap_hip (handle layout, device buffers, HIP error handling)
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <mutex>
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

template <typename T> struct DBuf {
    T *p = nullptr;
    size_t cap = 0;
    DBuf() = default;
    DBuf(const DBuf &) = delete;
    DBuf &operator=(const DBuf &) = delete;
    ~DBuf() {
        if (p) (void)hipFree(p);
    }
    void ensure(size_t n) {
        if (n <= cap && p) return;
        if (p) (void)hipFree(p);
        p = nullptr;
        size_t want = std::max<size_t>(n + n / 8, 64);
        hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
        if (e != hipSuccess) {
            p = nullptr;
            cap = 0;
            size_t fr = 0, tot = 0;
            (void)hipMemGetInfo(&fr, &tot);
            throw HipError("device scratch allocation of " + std::to_string(want * sizeof(T) >> 20) + " MB failed (" +
                           hipGetErrorString(e) + "; " + std::to_string(fr >> 20) + " MB free): use a smaller query batch");
        }
        cap = want;
    }
    // exactly n elements (the immutable index arrays: no head-room, optionally zero-filled)
    void alloc_exact(size_t n, bool zero = false, hipStream_t st = nullptr) {
        release();
        size_t want = std::max<size_t>(n, 1);
        HIPCHK(hipMalloc((void **)&p, want * sizeof(T)));
        cap = want;
        if (zero) HIPCHK(hipMemsetAsync(p, 0, want * sizeof(T), st));
    }
    void release() {
        if (p) (void)hipFree(p);
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

struct lm_index {
    lm::Work *work = nullptr;       // device scratch reused across calls (grow-only)
    lm::AlignCtx *actx[2] = {nullptr, nullptr}; // one per alignment worker
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
    DBuf<uint8_t> tmp, tmp2; // rocPRIM temporary storage (per stream)
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
};

