// lm_pipeline.hip — host orchestration of the batched `lexicmap search` pipeline and the C-ABI (include/lexicmap_hip.h).
//
// The host does bookkeeping only: buffer management, rocPRIM sorts/scans between kernels, the contig/coordinate glue of
// lib-index-search.go:2083-2468, BLAST statistics (lib-index-search-util.go:260-304) and result ordering
// (:2701-2749,2919-2932).  All of §8(a) rows a1-a12,a14,a15 run in the kernels of lm_kernels.hip.  There is no CPU path:
// without a HIP device every entry point fails with LM_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <malloc.h>
#include <cstring>
#include <map>
#include <mutex>
#include <sys/stat.h>
#include <stdexcept>
#include <tuple>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/lexicmap_hip.h"
#include "lm_format.h"
#include "lm_kernels.h"

#include "lm_internal.h"
#include "lm_prims.h"
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <future>
#include <memory>
#include <exception>
#include <list>
#include <thread>

thread_local std::string g_open_error;

namespace lm {

// ---- profiling: HIP events on the library's stream around each named launch ---------------------------------
// Helpers take the stream and the rocPRIM scratch from here: a host thread may install its own (thread-local) pair to
// run part of a batch beside the handle's stream; by default it is the handle's.
static thread_local DBuf<uint8_t> *tls_tmp = nullptr;
static thread_local int tls_lane = 0; // which of the handle's two lanes this thread works for
static inline Work *&lane_work(lm_index *ix) { return tls_lane ? ix->work1 : ix->work; }
static inline AlignCtx **lane_actx(lm_index *ix) { return tls_lane ? ix->actx1 : ix->actx; }
static inline hipStream_t &lane_st(lm_index *ix) { return tls_lane ? ix->st_b : ix->st; }
static inline hipStream_t &lane_st2(lm_index *ix) { return tls_lane ? ix->st2_b : ix->st2; }
static inline DBuf<uint8_t> &lane_tmp(lm_index *ix) { return tls_lane ? ix->tmp_b : ix->tmp; }
static inline DBuf<uint8_t> &lane_tmp2(lm_index *ix) { return tls_lane ? ix->tmp2_b : ix->tmp2; }
// the scratch budget of the lane this thread works for
static inline int64_t BUDGET(lm_index *ix) { return ix->scratch_budget > 0 ? ix->scratch_budget / ix->budget_lanes : ix->scratch_budget; }
static int device_cus(int device) {
    static int cus[64] = {0};
    if (device < 0 || device >= 64) return 256;
    if (!cus[device]) {
        hipDeviceProp_t p;
        cus[device] = hipGetDeviceProperties(&p, device) == hipSuccess && p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    }
    return cus[device];
}
static double g_dbg_t0 = 0; // LM_DEBUG: start of the running search (time stamps of the debug lines)
static inline void dbg_stamp(const char *what) {
    if (getenv("LM_DEBUG")) fprintf(stderr, "[lm +%.1f ms] %s\n", now_ms() - g_dbg_t0, what);
}
static inline hipStream_t S(lm_index *ix) { return tls_stream ? tls_stream : lane_st(ix); }
static inline DBuf<uint8_t> &TMP(lm_index *ix) { return tls_tmp ? *tls_tmp : lane_tmp(ix); }

struct Prof {
    lm_index *ix;
    int entry = -1;
    hipEvent_t a = nullptr, b = nullptr;
    Prof(lm_index *ix_, const char *name, int64_t bytes = 0) : ix(ix_) {
        if (!ix->prof) return;
        {
            std::lock_guard<std::mutex> l(ix->prof_mu);
            for (size_t i = 0; i < ix->prof_entries.size(); i++)
                if (ix->prof_entries[i].name == name) entry = (int)i;
            if (entry < 0) {
                ProfEntry e;
                e.name = name;
                ix->prof_entries.push_back(e);
                entry = (int)ix->prof_entries.size() - 1;
            }
            ix->prof_entries[entry].bytes += bytes;
            ix->prof_entries[entry].launches++;
        }
        HIPCHK(hipEventCreate(&a));
        HIPCHK(hipEventCreate(&b));
        HIPCHK(hipEventRecord(a, S(ix)));
    }
    ~Prof() {
        if (entry < 0) return;
        (void)hipEventRecord(b, S(ix));
        std::lock_guard<std::mutex> l(ix->prof_mu);
        ix->pending.push_back({entry, a, b});
    }
};

static void prof_add_bytes(lm_index *ix, const char *name, int64_t bytes) {
    if (!ix->prof) return;
    std::lock_guard<std::mutex> l(ix->prof_mu);
    for (auto &e : ix->prof_entries)
        if (e.name == name) e.bytes += bytes;
}

static void prof_resolve(lm_index *ix) {
    for (auto &p : ix->pending) {
        (void)hipEventSynchronize(p.b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, p.a, p.b);
        ix->prof_entries[p.entry].ms += ms;
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    ix->pending.clear();
}

// Go math.Log (pure Go / FreeBSD e_log.c) and math.Log2, for the gapScore table (lib-chaining.go:662-667):
// gapScore(g) = 0.1*g + 0.5*float32(math.Log2(float64(g))) in float32.
static double go_log(double x) {
    const double Ln2Hi = 6.93147180369123816490e-01, Ln2Lo = 1.90821492927058770002e-10, L1 = 6.666666666666735130e-01,
                 L2 = 3.999999999940941908e-01, L3 = 2.857142874366239149e-01, L4 = 2.222219843214978396e-01,
                 L5 = 1.818357216161805012e-01, L6 = 1.531383769920937332e-01, L7 = 1.479819860511658591e-01;
    int ki;
    double f1 = std::frexp(x, &ki);
    if (f1 < 0.70710678118654752440084436210484903928483593768847) {
        f1 *= 2;
        ki--;
    }
    double f = f1 - 1, k = (double)ki;
    double s = f / (2 + f), s2 = s * s, s4 = s2 * s2;
    double t1 = s2 * (L1 + s4 * (L3 + s4 * (L5 + s4 * L7)));
    double t2 = s4 * (L2 + s4 * (L4 + s4 * L6));
    double R = t1 + t2, hfsq = 0.5 * f * f;
    return k * Ln2Hi - ((hfsq - (s * (hfsq + R) + k * Ln2Lo)) - f);
}
static double go_log2(double x) {
    int e;
    double frac = std::frexp(x, &e);
    if (frac == 0.5) return (double)(e - 1);
    return go_log(frac) * 1.44269504088896340735992468100189214 + (double)e;
}
static float gap_score(float gap) {
    if (gap == 0) return 0;
    float a = 0.1f * gap;
    float b = 0.5f * (float)go_log2((double)gap);
    return a + b;
}

} // namespace lm

// gapScore table for integer gaps 0..ceil(max_gap) (shared with lm_builder.hip)
void lm_fill_gap_lut(lm_index *ix) {
    ix->gap_lut_n = (int)std::ceil(ix->opt.max_gap) + 2;
    std::vector<float> lut(ix->gap_lut_n);
    for (int g = 0; g < ix->gap_lut_n; g++) lut[g] = lm::gap_score((float)g);
    ix->d_gap_lut.ensure(lut.size());
    HIPCHK(hipMemcpyAsync(ix->d_gap_lut.p, lut.data(), lut.size() * sizeof(float), hipMemcpyHostToDevice, S(ix)));
    HIPCHK(hipStreamSynchronize(S(ix)));
}

// what is left of the device once the index image is resident bounds the per-batch scratch (query parts, WFA pools)
void lm_set_scratch_budget(lm_index *ix) {
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) return;
    // (0.85 until round 5: the pools' head-room and the buffers outside the arena took the device to 99 %.  Round 6 tried 0.76:
    // the parts and therefore the demand stay the same - lane slabs + overflow slabs 148 GB instead of 145, the smaller slabs
    // push more into overflow slabs that are sized on a grid - and 6.5 GB were left free instead of 15.7: kept at 0.80.)
    ix->scratch_budget = (int64_t)((double)fr * 0.80);
    if (const char *e = getenv("LM_SCRATCH_BUDGET_MB")) ix->scratch_budget = std::max<int64_t>(64, atoll(e)) << 20;
    if (getenv("LM_DEBUG"))
        fprintf(stderr, "[lm] index resident: %.2f GB, device free %.2f of %.2f GB, scratch budget %.2f GB\n",
                (double)ix->hbm_bytes / 1e9, (double)fr / 1e9, (double)tot / 1e9, (double)ix->scratch_budget / 1e9);
    // A production-size index cuts its lane slabs NOW, as part of opening it (hipMalloc of ~100 GB clears pages for seconds:
    // 2.4 s of the first C3 step when it was done there); a small index (tests, stage calls, several shard handles on one
    // device) leaves it to its first search.
    if (ix->hbm_bytes >= ((int64_t)4 << 30)) {
        const double t0 = lm::now_ms();
        lm_reserve_lane_slabs(ix);
        if (getenv("LM_DEBUG")) fprintf(stderr, "[lm] the lane slabs were cut in %.0f ms (part of opening a production-size index)\n", lm::now_ms() - t0);
    }
}
void lm_reserve_lane_slabs(lm_index *ix) {
    if (ix->tune.arena_reserve_pct <= 0 || ix->lane_slabs.asked || ix->scratch_budget <= 0) return;
    int64_t want = ix->scratch_budget / 100 * ix->tune.arena_reserve_pct;
    if (ix->hbm_bytes < ((int64_t)4 << 30)) {
        // A small index (tests, stage calls, shard handles sharing a device, a host that embeds the library beside other users of
        // the GPU) reaches this at its first search, possibly long after it was opened: the budget is taken from what is free NOW
        // (another handle's slabs may have appeared since), and the slabs it keeps until it is closed are capped - 72 % of the
        // device held by every 60-kb test index starved whatever came next (a second process got a budget too small for one
        // chunk of pseudo-alignment anchors).  What a large batch needs beyond the cap comes as overflow slabs, handed back by trim().
        size_t fr = 0, tot = 0;
        if (!getenv("LM_SCRATCH_BUDGET_MB") && hipMemGetInfo(&fr, &tot) == hipSuccess)
            ix->scratch_budget = std::min<int64_t>(ix->scratch_budget, (int64_t)((double)fr * 0.80));
        want = std::min<int64_t>(ix->scratch_budget / 100 * ix->tune.arena_reserve_pct, (int64_t)8 << 30);
    }
    const bool ok = ix->lane_slabs.reserve((size_t)want);
    if (getenv("LM_DEBUG") || getenv("LM_DEBUG_MEM"))
        fprintf(stderr, "[lm] scratch: %.2f GB of the %.2f-GB budget cut into two lane slabs: %s\n", (double)ix->lane_slabs.bytes() / 1e9,
                (double)ix->scratch_budget / 1e9, ok ? "yes" : "refused (slabs on demand)");
}

namespace lm {

template <typename T> static void h2d(lm_index *ix, DBuf<T> &d, const std::vector<T> &h) {
    d.ensure(std::max<size_t>(h.size(), 1));
    if (!h.empty()) HIPCHK(hipMemcpyAsync(d.p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, S(ix)));
}
template <typename T> static void d2h(lm_index *ix, std::vector<T> &h, const T *d, size_t n) {
    h.resize(n);
    if (n) HIPCHK(hipMemcpyAsync(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost, S(ix)));
}
static void sync(lm_index *ix) { HIPCHK(hipStreamSynchronize(S(ix))); }

// Big nested host containers (per-genome cluster/chain trees, task lists) are destroyed by a background thread: their
// destructors are hundreds of thousands of small frees that would otherwise sit between two batches with the GPU idle.
struct Janitor {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::function<void()>> q;
    bool stop = false;
    std::thread th;
    Janitor() : th([this] { run(); }) {}
    ~Janitor() {
        {
            std::lock_guard<std::mutex> l(mu);
            stop = true;
        }
        cv.notify_all();
        th.join();
    }
    void run() {
        std::unique_lock<std::mutex> l(mu);
        while (true) {
            cv.wait(l, [this] { return stop || !q.empty(); });
            if (q.empty() && stop) return;
            std::vector<std::function<void()>> work;
            work.swap(q);
            l.unlock();
            for (auto &f : work) f();
            work.clear();
            l.lock();
        }
    }
    template <typename T> void dispose(T &&obj) {
        auto p = std::make_shared<typename std::decay<T>::type>(std::move(obj));
        {
            std::lock_guard<std::mutex> l(mu);
            q.emplace_back([p]() mutable { p.reset(); });
        }
        cv.notify_one();
    }
};
static Janitor &janitor() {
    static Janitor j;
    return j;
}

// Every call moves ~150 MB through freshly allocated host vectors (tasks, chains, HSP records, rows). With glibc's
// defaults blocks above 128 KB are mmap'ed and unmapped again on free, so each call pays the page faults again (tens of
// ms per batch). Keep big blocks on the heap instead; LM_KEEP_MALLOC_DEFAULTS=1 opts out.
static void tune_malloc_once() {
    static bool done = [] {
        if (!getenv("LM_KEEP_MALLOC_DEFAULTS")) {
            mallopt(M_MMAP_THRESHOLD, 1 << 30);
            mallopt(M_TRIM_THRESHOLD, 2047 << 20);
            mallopt(M_TOP_PAD, 64 << 20);
        }
        return true;
    }();
    (void)done;
}

// Host-side glue between kernels is embarrassingly parallel over (query, genome) pairs: a small fork-join helper.
static int host_threads() { // cores this process may use: hardware threads capped by the cgroup CPU quota and by 24
    static int n = [] {
        int v = (int)std::max(1u, std::thread::hardware_concurrency());
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            long long quota = 0, period = 0;
            if (fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0)
                v = (int)std::min<long long>(v, std::max<long long>(1, quota / period));
            fclose(f);
        }
        if (const char *e = getenv("LM_HOST_THREADS")) { // several processes per host (one per GPU) share its cores
            int w = atoi(e);
            if (w >= 1) v = std::min(v, w);
        }
        return std::min(v, 24);
    }();
    return n;
}
template <typename F> static void parallel_for(int64_t n, int64_t grain, F f) {
    if (n <= 0) return;
    int64_t nchunks = (n + grain - 1) / grain;
    int nt = (int)std::min<int64_t>(nchunks, host_threads());
    if (nt <= 1) {
        f((int64_t)0, n);
        return;
    }
    std::atomic<int64_t> next{0};
    std::exception_ptr err;
    std::mutex emu;
    auto body = [&]() {
        try {
            while (true) {
                int64_t b = next.fetch_add(grain);
                if (b >= n) break;
                f(b, std::min(n, b + grain));
            }
        } catch (...) {
            std::lock_guard<std::mutex> l(emu);
            if (!err) err = std::current_exception();
        }
    };
    std::vector<std::thread> th;
    for (int i = 1; i < nt; i++) th.emplace_back(body);
    body();
    for (auto &t : th) t.join();
    if (err) std::rethrow_exception(err);
}

// exclusive scan of n+1 counts (counts[n] must be 0) into int64 offsets; returns the total after a sync
struct CastU32 {};
struct CastI32 {};
template <typename InT, typename Cast>
static int64_t scan_to_i64(lm_index *ix, const InT *counts, int64_t n, int64_t *offs) {
    prim_scan_to_i64(S(ix), TMP(ix), counts, (size_t)n, offs);
    int64_t total = 0;
    HIPCHK(hipMemcpyAsync(&total, offs + n, sizeof(int64_t), hipMemcpyDeviceToHost, S(ix)));
    sync(ix);
    return total;
}

static void sort_pairs_u64(lm_index *ix, uint64_t *k_in, uint64_t *k_out, uint64_t *v_in, uint64_t *v_out, int64_t n,
                           int begin_bit, int end_bit) {
    prim_sort_pairs(S(ix), TMP(ix), k_in, k_out, v_in, v_out, (size_t)n, begin_bit, end_bit);
}

static void sort_pairs_u32(lm_index *ix, uint32_t *k_in, uint32_t *k_out, uint32_t *v_in, uint32_t *v_out, int64_t n,
                           int begin_bit, int end_bit) {
    prim_sort_pairs(S(ix), TMP(ix), k_in, k_out, v_in, v_out, (size_t)n, begin_bit, end_bit);
}

// sort anchors by (A, B): LSD — stable sort by B, then by A. Result lands in (A0,B0).
static void sort_anchors(lm_index *ix, uint64_t *A0, uint64_t *B0, uint64_t *A1, uint64_t *B1, int64_t n, int a_bits) {
    if (n <= 0) return;
    sort_pairs_u64(ix, B0, B1, A0, A1, n, 0, 64);
    sort_pairs_u64(ix, A1, A0, B1, B0, n, 0, a_bits);
}

// Same order as sort_anchors for keys whose TBegin < 2^tbits and QBegin < 2^qbits: B = QBegin:27 | (32-Len):6 | TBegin:29 |
// 2 flags has long runs of zero bits then, which the LSD passes skip. Three ping-pong sorts: the result lands in (A1, B1).
static void sort_anchors_fields(lm_index *ix, uint64_t *A0, uint64_t *B0, uint64_t *A1, uint64_t *B1, int64_t n, int a_bits,
                                int qbits, int tbits) {
    if (n <= 0) return;
    sort_pairs_u64(ix, B0, B1, A0, A1, n, 0, std::min(31, 2 + tbits));
    sort_pairs_u64(ix, B1, B0, A1, A0, n, 31, std::min(64, 37 + qbits));
    sort_pairs_u64(ix, A0, A1, B0, B1, n, 0, a_bits);
}

// ---- query batch ---------------------------------------------------------------------------------------------
} // namespace lm

struct lm_qbatch {
    lm_index *ix = nullptr;
    int nq = 0;
    // a batch too large for one pass of the pipeline (device scratch, 32-bit slot numbers) is held as consecutive parts;
    // `parts` is empty for a plain batch. q0 = number of the part's first query in the caller's batch.
    std::vector<lm_qbatch *> parts;
    uint32_t q0 = 0;
    ~lm_qbatch() {
        for (auto *p : parts) delete p;
    }
    std::vector<uint8_t> h_seq;
    std::vector<int64_t> h_qoff, h_posoff;
    int64_t total_len = 0, total_pos = 0;
    DBuf<uint8_t> d_seq;
    DBuf<int64_t> d_qoff, d_posoff, d_segoff;
    // per-query prefix filter of the pseudo-alignment (k_build_cmp_bits): 2^bits_log[q] bits at word bits_off[q]
    DBuf<int64_t> d_bits_off;
    DBuf<int32_t> d_bits_log;
    int64_t bits_words = 0;
    // per-query bucket table over its sorted comparison k-mers (k_build_cmp_tab): 2^tab_bits[q] + 1 entries at tab_off[q]
    DBuf<int64_t> d_tab_off;
    DBuf<int32_t> d_tab_bits;
    int64_t tab_words = 0;
};

struct lm_result {
    std::vector<lm_hsp> rows;
    std::vector<std::string *> strings; // owned cigar/qseq/sseq/align
    lm_stage_stats stats;
    ~lm_result() {
        for (auto *s : strings) delete s;
    }
};

struct lm_stage {
    std::vector<uint32_t> u32a;
    std::vector<float> f32a;
    std::vector<uint64_t> u64a, u64b;
    std::vector<int64_t> i64a, i64b;
    std::vector<int32_t> i32a, i32b;
    std::vector<lm_pair> pairs;
    std::vector<lm_anchor> raw, cleared;
    std::vector<lm_chain2> chains2;
    std::vector<lm_wfa> wfa;
};

namespace lm {

// everything a batch keeps on the device between stages
struct Work {
    lm_index *ix;
    lm_qbatch *qb;
    // stage A
    DBuf<uint64_t> keys_all, keys_all2, keys_cmp, keys_cmp2;
    DBuf<uint32_t> vals_all, vals_all2, vals_cmp, vals_cmp2, first_mask, cmp_tab, cmp_bits;
    DBuf<int32_t> nvalid;
    uint64_t *k_all = nullptr, *k_cmp = nullptr; // sorted
    uint32_t *v_all = nullptr, *v_cmp = nullptr;
    // stage B
    DBuf<uint64_t> kmers;
    DBuf<int64_t> klo, khi;
    // stage C
    DBuf<uint32_t> lk_counts, lk_list, lk_list2, lk_perm, lk_perm2;
    DBuf<int64_t> lk_offs, lk_starts;
    DBuf<int32_t> lk_nscan;
    DBuf<uint64_t> lk_key, lk_rec; // per issued lookup, sorted order: the looked-up k-mer, (first location index << 32 | locations)
    DBuf<unsigned long long> stat;
    DBuf<uint64_t> A0, B0, A1, B1;
    int64_t n_anchors = 0;
    DBuf<uint64_t> segA;
    DBuf<int32_t> seg_len;
    DBuf<int64_t> seg_off;
    DBuf<int32_t> nseg_d;
    int nseg = 0;
    // stage D
    DBuf<LmSub> subs;
    DBuf<uint8_t> marks, visited;
    DBuf<uint64_t> msi, s2i;
    DBuf<int8_t> dirs;
    DBuf<int32_t> chain_off_pool, chain_idx_pool, seg_n, seg_nch, order_scratch, chain_big;
    DBuf<float> seg_score;
    // stage E
    DBuf<int32_t> ntask;
    DBuf<int64_t> task_off;
    DBuf<uint8_t> keep;
    DBuf<Task> tasks;
    PBuf<Task> tasks_host;
    DBuf<int32_t> task_wlen;
    DBuf<int64_t> task_woff;
    int64_t ntasks = 0;
    explicit Work(lm_index *i, lm_qbatch *q) : ix(i), qb(q) {
        for_each_seeding([](auto &b) { b.phase = true; });
    }
    // The seeding half's arrays (unsorted k-mer copies, captures, lookups, anchors, chaining scratch) are dead once the
    // tasks exist; the alignment half needs the room when they are large (long-read batches: tens of GB).  They are
    // "phase" buffers: the large ones live in the handle's scratch arena (lm_internal.h) and go back to it here, so the
    // alignment half is carved from the same slabs; small ones stay plain grow-only allocations.
    template <class F> void for_each_seeding(F f) {
        f(keys_all); f(keys_all2); f(vals_all); f(vals_all2); f(first_mask); f(kmers); f(klo); f(khi); f(lk_counts);
        f(lk_list); f(lk_list2); f(lk_perm); f(lk_perm2); f(lk_offs); f(lk_starts); f(lk_nscan); f(lk_key); f(lk_rec); f(A0); f(B0); f(A1);
        f(B1); f(segA); f(seg_len); f(seg_off); f(subs); f(marks); f(visited); f(msi); f(s2i); f(dirs);
        f(chain_off_pool); f(chain_idx_pool); f(seg_n); f(seg_nch); f(order_scratch); f(chain_big); f(seg_score); f(ntask);
        f(task_off); f(task_wlen); f(task_woff);
    }
    void release_seeding(int64_t keep_below_bytes) {
        auto drop = [&](auto &b) {
            if (b.arena || (int64_t)b.bytes() > keep_below_bytes) b.release();
        };
        // of the two buffers of the sorted comparison arrays the one holding the result survives (pseudo-alignment)
        if (k_cmp == keys_cmp2.p) drop(keys_cmp); else drop(keys_cmp2);
        if (v_cmp == vals_cmp2.p) drop(vals_cmp); else drop(vals_cmp2);
        k_all = nullptr;
        v_all = nullptr;
        for_each_seeding(drop);
    }
    void rebind(lm_qbatch *q) {
        qb = q;
        n_anchors = 0;
        nseg = 0;
        ntasks = 0;
    }
};

static void stage_kmers(Work &w) {
    lm_index *ix = w.ix;
    lm_qbatch *qb = w.qb;
    int64_t P = qb->total_pos;
    size_t n2 = (size_t)std::max<int64_t>(2 * P, 1);
    w.keys_all.ensure(n2);
    w.keys_all2.ensure(n2);
    w.keys_cmp.ensure(n2);
    w.keys_cmp2.ensure(n2);
    w.vals_all.ensure(n2);
    w.vals_all2.ensure(n2);
    w.vals_cmp.ensure(n2);
    w.vals_cmp2.ensure(n2);
    w.nvalid.ensure(qb->nq + 1);
    HIPCHK(hipMemsetAsync(w.nvalid.p, 0, sizeof(int32_t) * (qb->nq + 1), S(ix)));
    w.k_all = w.keys_all.p;
    w.v_all = w.vals_all.p;
    w.k_cmp = w.keys_cmp.p;
    w.v_cmp = w.vals_cmp.p;
    if (P == 0) return;
    {
        Prof p(ix, "k_extract_kmers", qb->total_len + 2 * P * 24);
        launch_extract_kmers(S(ix), qb->d_seq.p, qb->d_qoff.p, qb->d_posoff.p, qb->nq, ix->host.k, P, w.keys_all.p,
                             w.vals_all.p, w.keys_cmp.p, w.vals_cmp.p, w.nvalid.p);
    }
    {
        Prof p(ix, "segmented_sort_kmers", 2 * (2 * P) * 24); // two sorts of 2P (u64 key, u32 value) pairs, in and out
        int end_bit = std::min(64, 2 * ix->host.k);
        prim_segmented_sort_pairs(S(ix), TMP(ix), w.keys_all.p, w.keys_all2.p, w.vals_all.p, w.vals_all2.p, (size_t)(2 * P),
                                  (size_t)qb->nq, qb->d_segoff.p, 0, end_bit);
        prim_segmented_sort_pairs(S(ix), TMP(ix), w.keys_cmp.p, w.keys_cmp2.p, w.vals_cmp.p, w.vals_cmp2.p, (size_t)(2 * P),
                                  (size_t)qb->nq, qb->d_segoff.p, 0, 64);
    }
    w.k_all = w.keys_all2.p;
    w.v_all = w.vals_all2.p;
    w.k_cmp = w.keys_cmp2.p;
    w.v_cmp = w.vals_cmp2.p;
    w.cmp_tab.ensure((size_t)qb->tab_words + 1);
    launch_build_cmp_tab(S(ix), w.k_cmp, qb->d_posoff.p, w.nvalid.p, qb->nq, ix->host.k, qb->d_tab_off.p, qb->d_tab_bits.p,
                         qb->tab_words, w.cmp_tab.p);
    w.cmp_bits.ensure((size_t)qb->bits_words + 1);
    HIPCHK(hipMemsetAsync(w.cmp_bits.p, 0, (size_t)qb->bits_words * sizeof(uint32_t), S(ix)));
    launch_build_cmp_bits(S(ix), w.k_cmp, qb->d_posoff.p, w.nvalid.p, qb->nq, ix->host.k, qb->d_bits_off.p, qb->d_bits_log.p,
                          w.cmp_bits.p);
}

static void stage_mask(Work &w) {
    lm_index *ix = w.ix;
    lm_qbatch *qb = w.qb;
    int M = ix->host.M;
    int64_t nqm = (int64_t)qb->nq * M;
    w.kmers.ensure((size_t)nqm + 1);
    w.klo.ensure((size_t)nqm + 1);
    w.khi.ensure((size_t)nqm + 1);
    w.first_mask.ensure((size_t)std::max<int64_t>(2 * qb->total_pos, 1));
    launch_fill_u32(S(ix), w.first_mask.p, 2 * qb->total_pos, 0xffffffffu);
    Prof p(ix, "k_mask", nqm * 32);
    launch_mask(S(ix), w.k_all, qb->d_posoff.p, qb->nq, M, ix->host.k, ix->view.masks, w.kmers.p, w.klo.p, w.khi.p,
                w.first_mask.p);
}

struct PartTooLarge : std::exception {
    double over = 2.0; // how many times the part's seed anchors exceed what one pass may hold
};
struct ChunkTooLarge : std::exception {};

static void stage_lookup(Work &w, lm_stage_stats &stats) {
    lm_index *ix = w.ix;
    lm_qbatch *qb = w.qb;
    int M = ix->host.M;
    int64_t nqm = (int64_t)qb->nq * M;
    int64_t n = nqm * 2; // (query, mask, direction) slots; the slot number travels as a u32
    if (n >= ((int64_t)1 << 32)) throw HipError("too many seed lookups in one batch; use a smaller query batch");
    w.stat.ensure(4);
    HIPCHK(hipMemsetAsync(w.stat.p, 0, 4 * sizeof(unsigned long long), S(ix)));
    // issued lookups, compacted: (sort key = list | partition, slot)
    w.lk_list.ensure((size_t)n);
    w.lk_list2.ensure((size_t)n);
    w.lk_perm.ensure((size_t)n);
    w.lk_perm2.ensure((size_t)n);
    {
        Prof p(ix, "k_lookup_prep", n * 12);
        launch_lookup_prep(S(ix), ix->view, w.kmers.p, w.klo.p, w.first_mask.p, nqm, w.lk_list.p, w.lk_perm.p, w.stat.p + 2);
    }
    unsigned long long nlk_u = 0;
    HIPCHK(hipMemcpyAsync(&nlk_u, w.stat.p + 2, sizeof nlk_u, hipMemcpyDeviceToHost, S(ix)));
    sync(ix);
    const int64_t nlk = (int64_t)nlk_u;
    stats.seed_lookups += nlk;
    w.n_anchors = 0;
    w.nseg = 0;
    if (nlk == 0) return;
    {
        // in (list, partition) order: neighbouring threads read neighbouring table rows and partitions; outlier
        // lookups (bit 31) sort to the end
        Prof p(ix, "sort_lookups", nlk * 16); // (u32 key + u32 slot) in and out
        sort_pairs_u32(ix, w.lk_list.p, w.lk_list2.p, w.lk_perm.p, w.lk_perm2.p, nlk, 0, 32);
    }
    w.lk_counts.ensure((size_t)nlk + 1);
    w.lk_offs.ensure((size_t)nlk + 1);
    w.lk_starts.ensure((size_t)nlk + 1);
    w.lk_nscan.ensure((size_t)nlk + 1);
    w.lk_key.ensure((size_t)nlk + 1);
    w.lk_rec.ensure((size_t)nlk + 1);
    HIPCHK(hipMemsetAsync(w.lk_counts.p + nlk, 0, sizeof(uint32_t), S(ix)));
    {
        Prof p(ix, "k_lookup_count");
        launch_lookup_count(S(ix), ix->view, w.kmers.p, w.klo.p, w.khi.p, w.lk_list2.p, w.lk_perm2.p, nlk, ix->opt.min_prefix,
                            w.lk_counts.p, w.lk_starts.p, w.lk_nscan.p, w.stat.p, w.lk_key.p, w.lk_rec.p);
    }
    int64_t T;
    {
        Prof p(ix, "scan");
        T = scan_to_i64<uint32_t, CastU32>(ix, w.lk_counts.p, nlk, w.lk_offs.p);
    }
    unsigned long long hv = 0;
    HIPCHK(hipMemcpyAsync(&hv, w.stat.p, sizeof hv, hipMemcpyDeviceToHost, S(ix)));
    sync(ix);
    stats.seed_values += (int64_t)hv;
    // anchors + their chaining scratch (~96 B each) must fit 22 % of the scratch budget, and their number 31 bits:
    // otherwise the caller halves this part of the batch and comes back (lm_search_resident)
    const char *dbg_max = getenv("LM_DEBUG_MAX_ANCHORS"); // test hook: forces the halving path
    if (T >= (int64_t)1 << 31 || (BUDGET(ix) > 0 && T * 96 > BUDGET(ix) * 22 / 100) ||
        (dbg_max && qb->nq > 1 && T > atoll(dbg_max))) {
        if (qb->nq <= 1) throw HipError("one query yields more seed anchors than the device can hold");
        PartTooLarge e;
        e.over = std::max(T >= (int64_t)1 << 31 ? (double)T / 2147483647.0 : 1.0, BUDGET(ix) > 0 ? (double)T * 96.0 / ((double)BUDGET(ix) * 0.22) : 1.0);
        if (dbg_max) e.over = std::max(e.over, (double)T / (double)std::max<long long>(1, atoll(dbg_max)));
        throw e;
    }
    {   // (only for a part that goes on: a part thrown back for halving is searched again as two)
        // algorithmic bytes of the lookup, SURVEY.md §8(d) as written (reference-format sizes): per ISSUED lookup one
        // 8-byte anchor-table entry + ceil(log2 S_b) 8-byte k-mer probes, S_b = mean seeds per anchor partition, plus
        // 16 B (k-mer + value) per returned seed.  The packed image moves less than that: 2 x 4 B of table, key_bits / 8
        // per probe and (key_bits + val_bits) / 8 per returned seed (reported as k_lookup_count_packed).
        const double sb = (double)ix->n_seeds / std::max<double>(1.0, 2.0 * M * (double)(ix->view.P1 - 1));
        const int64_t probes = (int64_t)std::ceil(std::log2(sb + 1.0));
        prof_add_bytes(ix, "k_lookup_count", nlk * (8 + 8 * probes) + 16 * (int64_t)hv);
    }
    stats.anchors_raw += T;
    w.n_anchors = T;
    if (T == 0) return;
    w.A0.ensure((size_t)T);
    w.B0.ensure((size_t)T);
    w.A1.ensure((size_t)T);
    w.B1.ensure((size_t)T);
    {
        Prof p(ix, "k_lookup_emit", T * 16);
        if (ix->view.g_keep || !ix->tune.lookup_flat) // a genome whitelist: the kept seeds of a lookup are not a prefix of its range
            launch_lookup_emit(S(ix), ix->view, w.kmers.p, w.klo.p, w.khi.p, w.v_all, w.lk_list2.p, w.lk_perm2.p, nlk, w.lk_counts.p,
                               w.lk_offs.p, w.lk_starts.p, w.lk_nscan.p, w.A0.p, w.B0.p);
        else
            launch_lookup_emit_flat(S(ix), ix->view, w.v_all, w.lk_list2.p, w.lk_perm2.p, nlk, w.lk_counts.p, w.lk_offs.p, w.lk_starts.p,
                                    w.lk_key.p, w.lk_rec.p, w.A0.p, w.B0.p);
    }
    {
        Prof p(ix, "sort_anchors", T * 32); // two u64 per anchor in and out
        sort_anchors(ix, w.A0.p, w.B0.p, w.A1.p, w.B1.p, T, 64);
    }
    // segments = runs of equal A
    w.segA.ensure((size_t)T + 1);
    w.seg_len.ensure((size_t)T + 2);
    w.nseg_d.ensure(2);
    {
        Prof p(ix, "rle", T * 8);
        prim_rle(S(ix), TMP(ix), w.A0.p, (size_t)T, w.segA.p, w.seg_len.p, w.nseg_d.p);
    }
    int32_t nseg = 0;
    HIPCHK(hipMemcpyAsync(&nseg, w.nseg_d.p, sizeof nseg, hipMemcpyDeviceToHost, S(ix)));
    sync(ix);
    w.nseg = nseg;
    HIPCHK(hipMemsetAsync(w.seg_len.p + nseg, 0, sizeof(int32_t), S(ix)));
    w.seg_off.ensure((size_t)nseg + 2);
    scan_to_i64<int32_t, CastI32>(ix, w.seg_len.p, nseg, w.seg_off.p);
    stats.genome_pairs += nseg;
}

static LmChainOpt chain_opt(lm_index *ix) {
    LmChainOpt o;
    o.max_gap = (float)ix->opt.max_gap;
    o.min_score = lm_seed_weight((float)(uint8_t)ix->opt.min_single_prefix); // lib-index-search.go:742
    o.max_distance = (float)ix->opt.max_distance;
    o.top_chains = ix->opt.top_n_chains;
    o.gap_lut = ix->d_gap_lut.p;
    o.gap_lut_n = ix->gap_lut_n;
    return o;
}

static void stage_chain1(Work &w) {
    lm_index *ix = w.ix;
    int64_t T = w.n_anchors;
    int nseg = w.nseg;
    if (nseg == 0) return;
    w.subs.ensure((size_t)T);
    w.marks.ensure((size_t)T);
    w.visited.ensure((size_t)T);
    w.msi.ensure((size_t)T);
    w.s2i.ensure((size_t)T);
    w.dirs.ensure((size_t)T);
    w.chain_off_pool.ensure((size_t)T + 4 * (size_t)nseg + 8);
    w.chain_idx_pool.ensure(2 * (size_t)T + 8 * (size_t)nseg + 8);
    w.order_scratch.ensure((size_t)T + 4 * (size_t)nseg + 8);
    w.seg_n.ensure(nseg);
    w.seg_nch.ensure((size_t)nseg + 1);
    w.seg_score.ensure(nseg);
    // pairs with many anchors go to the wave-cooperative kernel (list built by the lane kernel)
    w.chain_big.ensure((size_t)nseg + 2);
    HIPCHK(hipMemsetAsync(w.chain_big.p, 0, sizeof(int32_t) * 2, S(ix))); // [0] = count, list from [2]
    Prof p(ix, "k_chain1", T * 8 * 3);
    launch_chain1(S(ix), w.B0.p, w.seg_off.p, nseg, chain_opt(ix), ix->host.k, w.subs.p, w.marks.p, w.msi.p, w.s2i.p,
                  w.dirs.p, w.visited.p, w.chain_off_pool.p, w.chain_idx_pool.p, w.seg_n.p, w.seg_score.p, w.seg_nch.p,
                  ix->tune.chain1_wave ? w.chain_big.p + 2 : nullptr, (unsigned int *)w.chain_big.p);
}

// ---- host-side result assembly types ---------------------------------------------------------------------------
struct HChain { // Chain2Result (lib-chaining2.go:106-135) as it moves through falin
    int qbegin, qend, tbegin, tend;
    int aligned_bases_q, matched_bases, aligned_length = 0, gaps = 0;
    double pident, aligned_fraction = 0, evalue = 0;
    int score = 0, bitscore = 0;
    int max_ext_len = 0, tpos_offset_begin = 0;
    bool alive = true;
    int64_t hsp = -1; // index into the HSP arrays of the current chunk
    std::string *cigar = nullptr, *qseq = nullptr, *tseq = nullptr, *align = nullptr;
};
struct HCluster { // SimilarityDetail (:1099-1120)
    bool rc = false, variant_a = false, has_result = false;
    int nseeds = 0, seq_idx = 0;
    int g = -1;                     // local genome record of this cluster (differs from the result's after a chunk merge)
    int nchunks = 1, chunk_idx = 0; // :1118-1119
    double sim = 0;
    int tBegin = 0, tEnd = 0; // chain window
    int64_t task = -1;
    std::vector<HChain> chains;
};
struct HGenome { // SearchResult (:1023-1040)
    uint32_t q = 0;
    uint64_t bg = 0;
    int g = -1;
    std::vector<HCluster> sds;
    double aligned_fraction = 0;
    bool alive = true;
};

struct AKey {
    int qb, qe, tb, te, seq, rc;
    bool operator<(const AKey &o) const {
        return std::tie(qb, qe, tb, te, seq, rc) < std::tie(o.qb, o.qe, o.tb, o.te, o.seq, o.rc);
    }
};

// lib-seq_compare.go:270-308
static int coverage_len(std::vector<std::pair<int, int>> &r) {
    if (r.empty()) return 0;
    if (r.size() == 1) return r[0].second - r[0].first + 1;
    std::stable_sort(r.begin(), r.end(), [](const std::pair<int, int> &a, const std::pair<int, int> &b) { return a.first < b.first; });
    int tot = 0, start = r[0].first, end = r[0].second;
    for (size_t i = 1; i < r.size(); i++) {
        if (r[i].first > end) {
            tot += end - start + 1;
            start = r[i].first;
            end = r[i].second;
            continue;
        }
        if (r[i].second <= end) continue;
        end = r[i].second;
    }
    return tot + end - start + 1;
}

// The contig-resolution / coordinate-conversion / dedup glue of falin for ONE lexichash chain (task) of a genome:
// lib-index-search.go:2079-2470. Appends clusters (with pre-WFA HSPs) to `gen`.
static void glue_task(const lm_index *ix, HGenome &gen, std::map<AKey, bool> &keys, const Task &t, int64_t task_id,
                      const LmChain2 *cr, int ncr) {
    if (ncr == 0) return;
    const HostGenome &G = ix->host.genomes[t.g];
    const int K = ix->host.k, contig_interval = ix->host.contig_interval;
    const bool rc = t.rc != 0;
    const int tBegin = t.tBegin, tEnd = t.tEnd, seqlen_w = t.wlen;
    int iSeqPre = -1, iSeq = 0, tPosOffsetBegin = 0, tPosOffsetEnd = 0;
    HCluster cur;
    auto new_cluster = [&](bool variant_a) {
        cur = HCluster();
        cur.rc = rc;
        cur.nseeds = t.nseeds;
        cur.tBegin = tBegin;
        cur.tEnd = tEnd;
        cur.task = task_id;
        cur.variant_a = variant_a;
        cur.g = t.g;
        if (ix->host.has_chunks) { // :2375-2385 / :2643-2653
            auto it = ix->host.chunk_of.find(t.bg);
            if (it != ix->host.chunk_of.end()) {
                cur.nchunks = it->second.n;
                cur.chunk_idx = it->second.idx;
            }
        }
    };
    new_cluster(false);
    auto convert = [&](HChain &c, int qb, int qe, int tb, int te, int iseq) {
        c.qbegin = qb;
        c.qend = qe;
        c.tpos_offset_begin = tPosOffsetBegin;
        if (rc) {
            c.tbegin = tBegin - tPosOffsetBegin + (seqlen_w - te - 1);
            if (c.tbegin < 0) {
                c.qend += c.tbegin;
                c.aligned_bases_q += c.tbegin;
                c.tbegin = 0;
            }
            c.tend = tBegin - tPosOffsetBegin + (seqlen_w - tb - 1);
            if (c.tend > G.seq_sizes[iseq] - 1) {
                c.qbegin += c.tend - (G.seq_sizes[iseq] - 1);
                c.tend = G.seq_sizes[iseq] - 1;
            }
        } else {
            c.tbegin = tBegin - tPosOffsetBegin + tb;
            if (c.tbegin < 0) {
                c.qbegin -= c.tbegin;
                c.aligned_bases_q += c.tbegin;
                c.tbegin = 0;
            }
            c.tend = tBegin - tPosOffsetBegin + te;
            if (c.tend > G.seq_sizes[iseq] - 1) {
                c.qend -= c.tend - (G.seq_sizes[iseq] - 1);
                c.tend = G.seq_sizes[iseq] - 1;
            }
        }
        c.max_ext_len = G.seq_sizes[iseq] - 1 - c.tend;
    };
    for (int _i = 0; _i < ncr; _i++) {
        const LmChain2 &s = cr[_i];
        HChain c;
        c.aligned_bases_q = s.aligned_bases_q;
        c.matched_bases = s.matched_bases;
        c.pident = s.pident;
        int qb = s.qbegin, qe = s.qend, tb = s.tbegin, te = s.tend;
        iSeq = 0;
        tPosOffsetBegin = 0;
        tPosOffsetEnd = 0;
        if (G.nseqs > 1) {
            iSeq = -1;
            int _begin, _end;
            if (rc) {
                _begin = tEnd - te + K;
                _end = tEnd - tb - K;
            } else {
                _begin = tBegin + tb + K;
                _end = tBegin + te - K;
            }
            if (_begin >= _end) {
                if (rc) {
                    _begin = tEnd - te;
                    _end = tEnd - tb;
                } else {
                    _begin = tBegin + tb;
                    _end = tBegin + te;
                }
            }
            for (int j = 0; j < G.nseqs; j++) {
                int l = G.seq_sizes[j];
                tPosOffsetEnd += l - 1;
                if (_begin + K >= tPosOffsetBegin && _end - K <= tPosOffsetEnd) {
                    iSeq = j;
                    break;
                } else if (_end < tPosOffsetBegin) {
                    iSeq = -1;
                    break;
                }
                tPosOffsetEnd += contig_interval + 1;
                tPosOffsetBegin = tPosOffsetEnd;
            }
            if (iSeq < 0) continue;
            if (iSeqPre >= 0 && iSeq != iSeqPre) { // the HSP fragment belongs to another contig (:2156-2415)
                int iSeq0 = iSeq;
                iSeq = iSeqPre;
                convert(c, qb, qe, tb, te, iSeq);
                if (!cur.chains.empty()) {
                    cur.variant_a = true;
                    cur.seq_idx = iSeq;
                    gen.sds.push_back(std::move(cur));
                }
                new_cluster(false);
                iSeqPre = -1;
                AKey key{c.qbegin, c.qend, c.tbegin, c.tend, iSeq, rc ? 1 : 0};
                if (!keys.count(key)) {
                    keys[key] = true;
                    cur.chains.push_back(c);
                }
                iSeq = iSeq0;
                continue;
            }
        }
        iSeqPre = iSeq;
        convert(c, qb, qe, tb, te, iSeq);
        AKey key{c.qbegin, c.qend, c.tbegin, c.tend, iSeq, rc ? 1 : 0};
        if (!keys.count(key)) {
            keys[key] = true;
            cur.chains.push_back(c);
        }
    }
    if (iSeq >= 0 && !cur.chains.empty()) {
        cur.seq_idx = iSeq;
        gen.sds.push_back(std::move(cur));
    }
}

} // namespace lm

// ================================================================================================================
// C-ABI
// ================================================================================================================
extern "C" {

void lm_options_default(lm_options *o) {
    memset(o, 0, sizeof *o);
    o->min_prefix = 15;
    o->min_single_prefix = 17;
    o->top_n_genomes = 0;
    o->top_n_chains = 0;
    o->max_gap = 50;
    o->max_distance = 1000;
    o->ext_len = 1000;
    o->ext_len2 = 50;
    o->min_qcov_per_genome = 0;
    o->max_evalue = 10;
    o->output_seq = 0;
    o->align_max_gap = 20;
    o->align_band = 100;
    o->align_min_match_len = 50;
    o->align_min_pident = 70;
    o->min_qcov_per_hsp = 0;
    o->shard_rank = 0;
    o->shard_count = 1;
    o->total_bases_override = 0;
}

const char *lm_last_error(const lm_index *idx) { return idx ? idx->err.c_str() : g_open_error.c_str(); }

static lm_status check_options(const lm_options &o, int k, int mask_prefix, int anchor_prefix, std::string &err) {
    // search.go:159-229 and lib-index-search.go:483-485
    if (o.min_prefix < 5 || o.min_prefix > 32) {
        err = "the value of flag -p/--seed-min-prefix should be in the range of [5, 32]";
        return LM_ERR_OPTION;
    }
    if (o.min_prefix > k || o.min_prefix < mask_prefix + anchor_prefix) {
        err = "MinPrefix (" + std::to_string(o.min_prefix) + ") should be in the range of [" +
              std::to_string(mask_prefix + anchor_prefix) + ", " + std::to_string(k) + "]";
        return LM_ERR_OPTION;
    }
    if (o.min_single_prefix < o.min_prefix || o.min_single_prefix > 32) {
        err = "the value of flag -P/--seed-min-single-prefix should be >= -p and <= 32";
        return LM_ERR_OPTION;
    }
    if (o.align_band < o.align_max_gap) {
        err = "the value of flag --align-band should be >= --align-max-gap";
        return LM_ERR_OPTION;
    }
    if (o.align_min_match_len < o.min_single_prefix) {
        err = "the value of flag -l/--align-min-match-len should be >= -P/--seed-min-single-prefix";
        return LM_ERR_OPTION;
    }
    if (o.align_min_pident < 60 || o.align_min_pident > 100) {
        err = "the value of flag -i/--align-min-match-pident should be in range of [60, 100]";
        return LM_ERR_OPTION;
    }
    if (o.max_gap <= 0 || o.max_distance <= 0 || o.ext_len < 0) {
        err = "seed-max-gap / seed-max-dist / align-ext-len out of range";
        return LM_ERR_OPTION;
    }
    return LM_OK;
}

void lm_index_close(lm_index *ix);
lm_status lm_index_open(const char *dir, const lm_options *opt, int device, lm_index **out) {
    *out = nullptr;
    g_open_error.clear();
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        g_open_error = "no HIP device available (this library has no CPU path)";
        return LM_ERR_NO_DEVICE;
    }
    lm_index *ix = new lm_index();
    try {
        ix->opt = *opt;
        ix->device = device;
        int status = 0;
        const double t_li0 = now_ms();
        // everything but the genome batches; those (names, contig tables, 2-bit bases: GBs) are read by a host thread BESIDE the
        // seed passes, which only need the genome count, the longest genome and the batch / shard tables
        std::string e = load_index(dir, opt->shard_count > 1 ? opt->shard_rank : 0,
                                   opt->shard_count > 1 ? opt->shard_count : 1, ix->host, status, false);
        if (getenv("LM_DEBUG")) fprintf(stderr, "[lm] loader: info, masks, batch and chunk tables read in %.0f ms\n", now_ms() - t_li0);
        if (!e.empty()) {
            g_open_error = e;
            lm_index_close(ix);
            return status == 2 ? LM_ERR_FORMAT : LM_ERR_IO;
        }
        if (opt->total_bases_override > 0) ix->host.total_bases = opt->total_bases_override;
        lm_status st = check_options(*opt, ix->host.k, ix->host.mask_prefix, ix->host.anchor_prefix, g_open_error);
        if (st != LM_OK) {
            lm_index_close(ix);
            return st;
        }
        HIPCHK(hipSetDevice(device));
        HIPCHK(hipStreamCreate(&ix->st));
        HostIndex &h = ix->host;
        // mask prefix table
        int p = h.mask_prefix;
        std::vector<int32_t> pfx((size_t)(1ull << (2 * p)) + 1, 0);
        for (int i = 0; i < h.M; i++) pfx[(size_t)(h.masks[i] >> ((h.k - p) << 1)) + 1]++;
        for (size_t i = 1; i < pfx.size(); i++) pfx[i] += pfx[i - 1];
        h2d(ix, ix->d_masks, h.masks);
        h2d(ix, ix->d_pfx_first, pfx);
        int gstatus = 0;
        const double t_g0 = now_ms();
        // The packed bases go from the reader's buffer - pinned, a run of records of ~256 MB at a time - straight to their place on
        // the device, on the reader's own stream: no host copy of the genome store and no buffer of a batch file's size (12.5 GB
        // at C2 size: cutting and faulting in 6-GB buffers, appending to the store and uploading it from pageable memory
        // afterwards was 8 s of a 9.7-s open once the seed passes took 4.3 s).
        ix->d_gbits.alloc_exact(h.gbits_bound + 64, true, S(ix)); // zero-filled: the padding behind every genome
        sync(ix);
        if (getenv("LM_DEBUG")) fprintf(stderr, "[lm] loader: the genome store (%.1f GB, zero-filled) was allocated in %.0f ms\n", (double)h.gbits_bound / 1e9, now_ms() - t_g0);
        struct GenomeSink {
            hipStream_t st = nullptr;
            uint8_t *pinned = nullptr; // the reader's buffer (grow-only)
            size_t pinned_cap = 0;
            uint8_t *dst = nullptr;
            size_t cap = 0;
            ~GenomeSink() {
                if (st) (void)hipStreamSynchronize(st);
                if (pinned) (void)hipHostFree(pinned);
                if (st) (void)hipStreamDestroy(st);
            }
        } gsink;
        gsink.dst = ix->d_gbits.p;
        gsink.cap = h.gbits_bound + 64;
        HIPCHK(hipStreamCreateWithFlags(&gsink.st, hipStreamNonBlocking));
        const int gdev = device;
        h.gbits_buffer = [&gsink, gdev](size_t bytes) -> uint8_t * {
            (void)hipSetDevice(gdev);
            if (bytes > gsink.pinned_cap) {
                if (gsink.pinned) {
                    (void)hipStreamSynchronize(gsink.st);
                    (void)hipHostFree(gsink.pinned);
                }
                gsink.pinned = nullptr;
                gsink.pinned_cap = 0;
                const size_t want = std::max<size_t>(bytes, (size_t)257 << 20);
                if (hipHostMalloc((void **)&gsink.pinned, want, hipHostMallocDefault) != hipSuccess) {
                    (void)hipGetLastError();
                    gsink.pinned = nullptr;
                    return nullptr;
                }
                gsink.pinned_cap = want;
            }
            return gsink.pinned;
        };
        h.gbits_sink = [&gsink](const uint8_t *src, size_t nbytes, int64_t off) {
            if (off < 0 || (size_t)off + nbytes + 16 > gsink.cap) return false;
            return hipMemcpyAsync(gsink.dst + off, src, nbytes, hipMemcpyHostToDevice, gsink.st) == hipSuccess;
        };
        h.gbits_batch_end = [&gsink]() { (void)hipStreamSynchronize(gsink.st); }; // the buffer is read into again
        std::future<std::string> gfut = std::async(std::launch::async, [&]() { return load_index_genomes(dir, h, gstatus); });
        const int64_t max_len = h.max_genome_len;
        h2d(ix, ix->d_batch_first, h.batch_first);
        if (!h.g2local.empty()) h2d(ix, ix->d_g2local, h.g2local);
        lm_fill_gap_lut(ix);
        sync(ix);
        DevIndexView &v = ix->view;
        v.g2local = h.g2local.empty() ? nullptr : ix->d_g2local.p;
        v.K = h.k;
        v.M = h.M;
        v.mask_prefix = h.mask_prefix;
        v.masks = ix->d_masks.p;
        v.pfx_first = ix->d_pfx_first.p;
        v.batch_first = ix->d_batch_first.p;
        v.nbatches = h.genome_batches;
        v.ngenomes = h.n_local_genomes;
        v.shard_rank = h.shard_rank;
        v.shard_count = h.shard_count;
        // (A device allocation that fails inside the passes - the keep budget is an estimate of what the packed image and its
        // sort scratch leave - makes the passes start over WITHOUT the kept copies, every file decoded twice, instead of failing the open.)
        bool loader_keep = getenv("LM_LOADER_NO_KEEP") == nullptr;
        for (int load_try = 0;; load_try++) try
        {   // packed seed image: every chunk file is decoded and shown to the packer twice (count, then place).  Files are
            // decoded by the host threads ahead of the upload into a small pool of REUSED chunk slots (after the first files
            // no decode touches fresh pages: faulting in and zero-filling 18 B per seed of new memory per file cost more than
            // the decoding itself), whose arrays are registered with the driver once they stop growing, so that the uploads
            // are DMA from pinned memory.  The host never holds more than the pool (the reference's RAM form of the whole
            // index would be 16 B/seed of host memory).
            SeedPacker sp;
            sp.begin(ix, h.n_local_genomes, max_len);
            const size_t nf = h.seed_files.size();
            size_t nslots = std::min<size_t>(std::max<size_t>(2, (size_t)host_threads()), std::max<size_t>(1, nf));
            int64_t biggest = 1;
            {   // a slot holds a file and its decoded seeds (~3.6 x the file): the pool may take a third of the free host memory
                for (auto &f : h.seed_files) {
                    struct stat sb;
                    if (stat(f.c_str(), &sb) == 0) biggest = std::max<int64_t>(biggest, (int64_t)sb.st_size);
                }
                int64_t avail = (int64_t)64 << 30;
                if (FILE *mf = fopen("/proc/meminfo", "r")) {
                    char line[256];
                    while (fgets(line, sizeof line, mf)) {
                        long long kb = 0;
                        if (sscanf(line, "MemAvailable: %lld kB", &kb) == 1) avail = kb * 1024;
                    }
                    fclose(mf);
                }
                const int64_t per_slot = biggest * 36 / 10 + (1 << 20);
                nslots = (size_t)std::max<int64_t>(std::min<int64_t>(2, (int64_t)nf), std::min<int64_t>((int64_t)nslots, avail / 3 / per_slot));
                if (nslots < 1) nslots = 1;
            }
            struct Slot {
                SeedChunk c;
                void *reg[3] = {nullptr, nullptr, nullptr}; // what is registered with the driver (the arrays' current storage)
                ~Slot() {
                    for (void *r : reg)
                        if (r) (void)hipHostUnregister(r);
                }
                void pin() { // (re-)register the arrays when their storage moved: they only grow, a few times in all
                    void *cur[3] = {c.kmers.data(), c.vals.data(), c.masks.data()};
                    const size_t bytes[3] = {c.kmers.size() * 8, c.vals.size() * 8, c.masks.size() * 2};
                    for (int j = 0; j < 3; j++) {
                        if (cur[j] == reg[j] || bytes[j] == 0) continue;
                        if (reg[j]) (void)hipHostUnregister(reg[j]);
                        reg[j] = hipHostRegister(cur[j], bytes[j], hipHostRegisterDefault) == hipSuccess ? cur[j] : nullptr;
                        if (!reg[j]) (void)hipGetLastError(); // pageable upload then: slower, not wrong
                    }
                }
            };
            DBuf<uint64_t> dk, dv;
            DBuf<uint16_t> dm;
            const bool pin = true; // (decoded chunks are registered with the driver: the upload is one DMA)
            const bool ldbg = getenv("LM_DEBUG") != nullptr;
            double t_wait = 0, t_pin = 0, t_pack = 0;
            const double t_seeds0 = now_ms();
            if (ldbg) fprintf(stderr, "[lm] loader: the seed passes start %.0f ms after the genome reader\n", t_seeds0 - t_g0);
            std::vector<std::unique_ptr<Slot>> slot(nslots); // (both passes use the same slots: warm pages, registered once)
            for (auto &sl : slot) {
                sl.reset(new Slot());
                // cut for the largest file by the slot's first decode (on its thread): the arrays never move afterwards, so a
                // registered range is never freed behind the driver's back
                sl->c.min_file_bytes = (size_t)biggest;
                sl->c.min_seeds = (size_t)biggest / 7 + 1;
            }
            // Decoded seeds that stay on the device between the two passes: a file whose (k-mer, value, mask) arrays - 18 bytes
            // per seed this shard keeps - fit beside what is still to be allocated (the packed image, the genomes) is decoded and
            // uploaded ONCE; its second pass reads the device copy.  A shard keeps 1 / N of the seeds: all of its files stay.
            struct Kept {
                DBuf<uint64_t> k, v;
                DBuf<uint16_t> m;
                int64_t n = 0;
            };
            std::vector<std::unique_ptr<Kept>> kept(nf);
            int64_t keep_budget = 0, kept_bytes = 0, files_bytes = 0;
            {
                for (auto &f : h.seed_files) {
                    struct stat sb;
                    if (stat(f.c_str(), &sb) == 0) files_bytes += (int64_t)sb.st_size;
                }
                size_t fr = 0, tot = 0;
                if (hipMemGetInfo(&fr, &tot) == hipSuccess) // (image: <= ~0.55 bytes per file byte measured; 0.75 reserved)
                    keep_budget = (int64_t)fr - files_bytes * 3 / 4 / (int64_t)std::max(1, h.shard_count) - ((int64_t)6 << 30); // (the genome store is allocated already)
                if (!loader_keep) keep_budget = 0;
            }
            int64_t n_kept_files = 0;
            for (int pass = 0; pass < 2; pass++) {
                // declaration order matters: `fut` is destroyed FIRST when an exception unwinds this scope (a failed upload,
                // DeviceOOM) and a std::async future joins its task in its destructor - so the decode tasks still running
                // have finished before the slots and status entries they write into are freed
                std::vector<size_t> files; // the files this pass decodes (pass 1: those without a device copy)
                for (size_t i = 0; i < nf; i++)
                    if (!kept[i]) files.push_back(i);
                const size_t nd = files.size();
                std::vector<int> stat(nf, 0), anch(nf, -1);
                std::vector<std::future<std::string>> fut(nd);
                auto launch = [&](size_t pos) {
                    const size_t i = files[pos];
                    fut[pos] = std::async(std::launch::async, [&, i, pos]() {
                        return decode_seed_chunk(h.seed_files[i], h, slot[pos % nslots]->c, stat[i], anch[i]);
                    });
                };
                for (size_t pos = 0; pos < std::min(nslots, nd); pos++) launch(pos);
                if (pass == 1) { // the files with a device copy, beside the decoders of the others
                    const double tk0 = now_ms();
                    for (size_t i = 0; i < nf; i++) {
                        if (!kept[i]) continue;
                        Kept &kp = *kept[i];
                        const int64_t slice = (int64_t)32 << 20;
                        for (int64_t o = 0; o < kp.n; o += slice) sp.place(kp.m.p + o, kp.k.p + o, kp.v.p + o, std::min(slice, kp.n - o));
                        sync(ix);
                        kept[i].reset(); // (its memory goes back before the partitions are sorted)
                    }
                    t_pack += now_ms() - tk0;
                }
                for (size_t pos = 0; pos < nd; pos++) {
                    const size_t i = files[pos];
                    const double tw0 = now_ms();
                    const std::string e2 = fut[pos].get();
                    t_wait += now_ms() - tw0;
                    if (!e2.empty()) {
                        for (size_t j = pos + 1; j < nd; j++)
                            if (fut[j].valid()) fut[j].wait();
                        g_open_error = e2;
                        const int stt = stat[i];
                        kept.clear();
                        slot.clear(); // (before the handle and its device context go)
                        gfut.wait();  // (the genome reader writes into the handle)
                        lm_index_close(ix);
                        return stt == 2 ? LM_ERR_FORMAT : LM_ERR_IO;
                    }
                    Slot &sl = *slot[pos % nslots];
                    const double tp0 = now_ms();
                    if (pin) sl.pin();
                    t_pin += now_ms() - tp0;
                    const double tk0 = now_ms();
                    SeedChunk &c = sl.c;
                    const int64_t cnt = (int64_t)c.n;
                    const int64_t slice = (int64_t)32 << 20;
                    Kept *kp = nullptr;
                    if (pass == 0 && cnt > 0 && kept_bytes + cnt * 18 <= keep_budget) {
                        std::unique_ptr<Kept> nk(new Kept());
                        try {
                            nk->k.alloc_exact((size_t)cnt);
                            nk->v.alloc_exact((size_t)cnt);
                            nk->m.alloc_exact((size_t)cnt);
                            nk->n = cnt;
                            kept[i] = std::move(nk);
                            kp = kept[i].get();
                            kept_bytes += cnt * 18;
                            n_kept_files++;
                        } catch (const std::exception &) { // the device said no: this file (and the rest) are decoded twice
                            (void)hipGetLastError();
                            keep_budget = 0;
                        }
                    }
                    for (int64_t o = 0; o < cnt; o += slice) {
                        const int64_t m = std::min(slice, cnt - o);
                        uint64_t *pk, *pv;
                        uint16_t *pm;
                        if (kp) {
                            pk = kp->k.p + o;
                            pv = kp->v.p + o;
                            pm = kp->m.p + o;
                        } else {
                            dk.ensure((size_t)m);
                            dv.ensure((size_t)m);
                            dm.ensure((size_t)m);
                            pk = dk.p;
                            pv = dv.p;
                            pm = dm.p;
                        }
                        HIPCHK(hipMemcpyAsync(pk, c.kmers.data() + o, (size_t)m * 8, hipMemcpyHostToDevice, S(ix)));
                        HIPCHK(hipMemcpyAsync(pv, c.vals.data() + o, (size_t)m * 8, hipMemcpyHostToDevice, S(ix)));
                        HIPCHK(hipMemcpyAsync(pm, c.masks.data() + o, (size_t)m * 2, hipMemcpyHostToDevice, S(ix)));
                        if (pass == 0)
                            sp.count(pm, pk, pv, m);
                        else
                            sp.place(pm, pk, pv, m);
                        if (!kp) sync(ix); // the staging buffers are reused
                    }
                    sync(ix); // (the slot's host arrays are decoded into again)
                    t_pack += now_ms() - tk0;
                    if (pos + nslots < nd) launch(pos + nslots); // this slot's next file
                }
                if (pass == 0) sp.end_count();
            }
            if (ldbg)
                fprintf(stderr, "[lm] loader: %lld of %zu chunk files kept on the device between the passes (%.2f GB of decoded seeds; budget %.2f GB)\n",
                        (long long)n_kept_files, nf, (double)kept_bytes / 1e9, (double)keep_budget / 1e9);
            const double tf0 = now_ms();
            if (load_try == 0 && loader_keep && getenv("LM_DEBUG_LOADER_OOM")) throw DeviceOOM("test hook: out of memory at the end of the seed passes");
            sp.finish();
            if (ldbg)
                fprintf(stderr, "[lm] loader: %zu chunk files through %zu slots, two passes in %.0f ms: %.0f ms waiting for the decoders, %.0f ms "
                                "registering the slots, %.0f ms uploading + packing, %.0f ms sorting the partitions\n",
                        nf, nslots, now_ms() - t_seeds0, t_wait, t_pin, t_pack, now_ms() - tf0);
            break;
        } catch (const DeviceOOM &e) {
            if (!loader_keep || load_try > 0) throw;
            loader_keep = false;
            (void)hipDeviceSynchronize();
            if (getenv("LM_DEBUG")) fprintf(stderr, "[lm] loader: %s - the seed passes start over without device copies of the decoded seeds\n", e.what());
        }
        // ---- the genomes (read meanwhile): bases and tables to the device
        {
            const std::string eg = gfut.get();
            if (getenv("LM_DEBUG")) fprintf(stderr, "[lm] loader: genome batches read %.0f ms after their reader started\n", now_ms() - t_g0);
            if (!eg.empty()) {
                g_open_error = eg;
                lm_index_close(ix);
                return gstatus == 2 ? LM_ERR_FORMAT : LM_ERR_IO;
            }
            if ((int64_t)h.genomes.size() != h.n_local_genomes) {
                g_open_error = "genome data: the batch files hold another number of genomes than their indexes";
                lm_index_close(ix);
                return LM_ERR_FORMAT;
            }
        }
        h.gbits_buffer = nullptr; // (they refer to this frame)
        h.gbits_sink = nullptr;
        h.gbits_batch_end = nullptr;
        std::vector<int64_t> goff;
        std::vector<int32_t> glen;
        std::vector<uint64_t> gbg;
        for (size_t i = 0; i < h.genomes.size(); i++) {
            goff.push_back(h.genomes[i].bits_off);
            glen.push_back(h.genomes[i].len);
            gbg.push_back(h.genomes[i].bg);
            ix->bg2local[h.genomes[i].bg] = (int)i;
        }
        h2d(ix, ix->d_g_off, goff);
        h2d(ix, ix->d_g_len, glen);
        h2d(ix, ix->d_g_bg, gbg);
        sync(ix);
        v.g_bg = ix->d_g_bg.p;
        v.gbits = ix->d_gbits.p;
        v.g_off = ix->d_g_off.p;
        v.g_len = ix->d_g_len.p;
        ix->hbm_bytes = ix->seed_bytes + (int64_t)(h.masks.size() * 8 + pfx.size() * 4 + (size_t)h.gbits_total + 64 + goff.size() * 20 +
                                                   h.batch_first.size() * 8);
        // the host copy of the packed genomes is no longer needed
        std::vector<uint8_t>().swap(h.gbits);
        ix->tmp.release();
        lm_set_scratch_budget(ix);
        if (getenv("LM_DEBUG")) fprintf(stderr, "[lm] loader: done %.0f ms after the genome reader started\n", now_ms() - t_g0);
    } catch (const std::exception &e) {
        g_open_error = e.what();
        lm_index_close(ix);
        return LM_ERR_HIP;
    }
    *out = ix;
    return LM_OK;
}

void lm_free_align_ctx(lm_index *ix, int lane = -1);

void lm_index_close(lm_index *ix) {
    if (!ix) return;
    prof_resolve(ix);
    delete ix->work;
    delete ix->work1;
    lm_free_align_ctx(ix); // AlignCtx is defined further down
    if (ix->st) (void)hipStreamDestroy(ix->st);
    if (ix->st2) (void)hipStreamDestroy(ix->st2);
    if (ix->st_b) (void)hipStreamDestroy(ix->st_b);
    if (ix->st2_b) (void)hipStreamDestroy(ix->st2_b);
    delete ix;
}

lm_status lm_index_get_info(const lm_index *ix, lm_index_info *info) {
    if (!ix || !info) return LM_ERR_ARG;
    info->k = ix->host.k;
    info->masks = ix->host.M;
    info->mask_prefix = ix->host.mask_prefix;
    info->anchor_prefix = ix->host.anchor_prefix;
    info->total_bases = ix->host.total_bases;
    info->genomes = (int64_t)ix->host.genomes.size();
    info->seeds = ix->n_seeds;
    int64_t gb = 0;
    for (auto &g : ix->host.genomes) gb += g.len;
    info->genome_bases = gb;
    info->hbm_bytes = ix->hbm_bytes;
    info->seed_bytes = ix->seed_bytes;
    info->outlier_seeds = ix->n_seeds_outlier;
    info->key_bits = ix->view.key_bits;
    info->val_bits = ix->view.gid_bits + ix->view.pos_bits + 1;
    info->partition_bases = ix->view.part_bases;
    info->pad = 0;
    return LM_OK;
}

const uint64_t *lm_index_masks(const lm_index *ix) { return ix ? ix->host.masks.data() : nullptr; }

void lm_profile_enable(lm_index *ix, int on) { ix->prof = on != 0; }
// measurement only: an empty kernel under a name of its own, launched when every stream of the device is idle.  bench.py puts one
// in front of and one behind the timed steps, so that tools/summarize_rocprof.py can restrict a rocprofv3 trace or counter pass
// to the dispatches of the (warm) timed steps - the passes of rounds 1-5 counted the cold first step of a fresh handle.
namespace lm {
__global__ void k_profile_mark(int) {}
} // namespace lm
void lm_profile_mark(lm_index *ix, int id) {
    if (!ix) return;
    std::lock_guard<std::mutex> lock(ix->mu);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(lm::k_profile_mark, dim3(1), dim3(64), 0, 0, id);
    (void)hipDeviceSynchronize();
}
// measurement switch (bench.py): exclusive != 0 serialises the searches that follow on this handle - no pseudo-alignment
// producer beside extend / WFA, the WFA length classes one after the other - so that the HIP-event time of a kernel is its
// own time and not that of whatever shared the chip with it.  Same results either way.
void lm_profile_exclusive(lm_index *ix, int exclusive) {
    if (!ix) return;
    std::lock_guard<std::mutex> lock(ix->mu);
    ix->tune.wfa_serial = exclusive != 0;
    ix->tune.no_pipeline = exclusive != 0;
}
void lm_tuning_reload(lm_index *ix) {
    if (!ix) return;
    std::lock_guard<std::mutex> lock(ix->mu);
    lm_tune fresh;
    fresh.wfa_serial = fresh.wfa_serial || ix->tune.wfa_serial;   // (owned by lm_profile_exclusive: a reload does not undo it)
    fresh.no_pipeline = fresh.no_pipeline || ix->tune.no_pipeline;
    ix->tune = fresh;
    // every variant starts from the same scratch state: both lanes' buffers go back (the lane slabs stay with the handle)
    (void)hipDeviceSynchronize();
    delete ix->work;
    ix->work = nullptr;
    delete ix->work1;
    ix->work1 = nullptr;
    lm_free_align_ctx(ix);
    ix->arena[0].trim();
    ix->arena[1].trim();
}
void lm_profile_reset(lm_index *ix) {
    prof_resolve(ix);
    ix->prof_entries.clear();
}
size_t lm_profile_get(lm_index *ix, const lm_kernel_time **out) {
    prof_resolve(ix);
    ix->prof_out.clear();
    for (auto &e : ix->prof_entries) ix->prof_out.push_back({e.name.c_str(), e.launches, e.ms, e.bytes});
    *out = ix->prof_out.data();
    return ix->prof_out.size();
}

// ---- query upload ---------------------------------------------------------------------------------------------
static lm_qbatch *upload_part(lm_index *ix, const lm_query *queries, size_t nq, uint32_t q0) {
    lm_qbatch *qb = new lm_qbatch();
    try {
        qb->ix = ix;
        qb->nq = (int)nq;
        qb->q0 = q0;
        qb->h_qoff.assign(nq + 1, 0);
        qb->h_posoff.assign(nq + 1, 0);
        int K = ix->host.k;
        for (size_t i = 0; i < nq; i++) {
            qb->h_qoff[i + 1] = qb->h_qoff[i] + queries[i].len;
            int64_t np = (int64_t)queries[i].len - K + 1;
            qb->h_posoff[i + 1] = qb->h_posoff[i] + (np > 0 ? np : 0);
        }
        qb->total_len = qb->h_qoff[nq];
        qb->total_pos = qb->h_posoff[nq];
        qb->h_seq.resize((size_t)qb->total_len + 64, 'A');
        for (size_t i = 0; i < nq; i++)
            if (queries[i].len) memcpy(&qb->h_seq[(size_t)qb->h_qoff[i]], queries[i].seq, queries[i].len);
        std::vector<int64_t> segoff(nq + 1);
        for (size_t i = 0; i <= nq; i++) segoff[i] = 2 * qb->h_posoff[i];
        h2d(ix, qb->d_seq, qb->h_seq);
        h2d(ix, qb->d_qoff, qb->h_qoff);
        h2d(ix, qb->d_posoff, qb->h_posoff);
        h2d(ix, qb->d_segoff, segoff);
        {   // prefix filters of the pseudo-alignment (lm_pa_bits_words, lm_algos.h): ~16 bits per k-mer (both strands) in
            // each hashed map, 2^13 .. 2^24 bits, + the Bloom filter and the exact 9-base map k_pa_filter keeps in LDS
            std::vector<int64_t> boff(nq + 1, 0);
            std::vector<int32_t> blog(nq + 1, 13);
            for (size_t i = 0; i < nq; i++) {
                const int64_t nk = 2 * (qb->h_posoff[i + 1] - qb->h_posoff[i]);
                int lg = 13;
                while (lg < 24 && ((int64_t)1 << lg) < 16 * nk) lg++;
                blog[i] = lg;
                boff[i + 1] = boff[i] + (int64_t)lm_pa_bits_words(lg);
            }
            qb->bits_words = boff[nq];
            std::vector<int64_t> toff(nq + 1, 0);
            std::vector<int32_t> tbits(nq + 1, LM_TAB_BITS_MIN);
            for (size_t i = 0; i < nq; i++) { // about two buckets per k-mer
                const int64_t nk = 2 * (qb->h_posoff[i + 1] - qb->h_posoff[i]);
                int tb = LM_TAB_BITS_MIN;
                while (tb < LM_TAB_BITS_MAX && ((int64_t)1 << tb) < 2 * nk) tb++;
                tbits[i] = tb;
                toff[i + 1] = toff[i] + ((int64_t)1 << tb) + 1;
            }
            qb->tab_words = toff[nq];
            h2d(ix, qb->d_tab_off, toff);
            h2d(ix, qb->d_tab_bits, tbits);
            h2d(ix, qb->d_bits_off, boff);
            h2d(ix, qb->d_bits_log, blog);
        }
        sync(ix);
    } catch (...) {
        delete qb;
        throw;
    }
    return qb;
}

// Limits of one pass: slot numbers (query, mask, direction) and k-mer numbers travel as 32-bit values, and the arrays
// sized by them must fit the device memory left beside the index (LM_MAX_PART_KMERS overrides: tests).
static void part_limits(lm_index *ix, int64_t *max_pos, int64_t *max_qm) {
    int64_t pos = ((int64_t)1 << 30) - 1, qm = ((int64_t)1 << 31) - 1;
    // ~104 B per k-mer position (two sorted k-mer arrays with their double buffers, capture marks) and ~80 B per
    // (query, mask) pair (captured k-mers, location ranges, lookup lists) may take 15 % of the scratch budget
    if (ix->scratch_budget > 0) {
        const int64_t b = ix->scratch_budget * 15 / 100; // shares of the scratch budget
        pos = std::min<int64_t>(pos, std::max<int64_t>(b / 2 / 104, 1 << 16));
        qm = std::min<int64_t>(qm, std::max<int64_t>(b / 2 / 80, (int64_t)ix->host.M));
    }
    if (const char *e = getenv("LM_MAX_PART_KMERS")) pos = std::max<int64_t>(1, std::min<int64_t>(pos, atoll(e)));
    *max_pos = pos;
    *max_qm = qm;
}

extern "C" lm_status lm_qbatch_upload(lm_index *ix, const lm_query *queries, size_t nq, lm_qbatch **out) {
    *out = nullptr;
    if (!ix) return LM_ERR_ARG;
    if (nq >= ((size_t)1 << 31)) {
        ix->err = "query batch too large; split the batch";
        return LM_ERR_ARG;
    }
    try {
        std::lock_guard<std::mutex> lock(ix->mu);
        HIPCHK(hipSetDevice(ix->device));
        int64_t max_pos, max_qm;
        part_limits(ix, &max_pos, &max_qm);
        const int K = ix->host.k, M = ix->host.M;
        // greedy split into consecutive parts within the limits (a single query above them is refused)
        std::vector<size_t> cuts{0};
        int64_t pos = 0, nqp = 0;
        for (size_t i = 0; i < nq; i++) {
            const int64_t np = std::max<int64_t>((int64_t)queries[i].len - K + 1, 0);
            if (np > max_pos) {
                ix->err = "a query of " + std::to_string(queries[i].len) + " bases exceeds what one pass can hold (" +
                          std::to_string(max_pos) + " k-mers)";
                return LM_ERR_ARG;
            }
            if (nqp > 0 && (pos + np > max_pos || (nqp + 1) * (int64_t)M > max_qm)) {
                cuts.push_back(i);
                pos = 0;
                nqp = 0;
            }
            pos += np;
            nqp++;
        }
        cuts.push_back(nq);
        if (cuts.size() == 2) {
            *out = upload_part(ix, queries, nq, 0);
            return LM_OK;
        }
        lm_qbatch *top = new lm_qbatch();
        top->ix = ix;
        top->nq = (int)nq;
        try {
            for (size_t c = 0; c + 1 < cuts.size(); c++)
                top->parts.push_back(upload_part(ix, queries + cuts[c], cuts[c + 1] - cuts[c], (uint32_t)cuts[c]));
        } catch (...) {
            delete top;
            throw;
        }
        for (auto *p : top->parts) {
            top->total_len += p->total_len;
            top->total_pos += p->total_pos;
        }
        *out = top;
        return LM_OK;
    } catch (const std::exception &e) {
        ix->err = e.what();
        return LM_ERR_HIP;
    }
}

void lm_qbatch_free(lm_qbatch *qb) { delete qb; }

} // extern "C"

namespace lm {

// ---- alignment half of the pipeline: tasks -> windows -> pseudo-alignment -> glue -> extend -> WFA -> finalize ----
struct TaskSpan { // a contiguous range of the host copy of the task list
    const Task *p = nullptr;
    size_t n = 0;
    const Task &operator[](size_t i) const { return p[i]; }
    size_t size() const { return n; }
    const Task &back() const { return p[n - 1]; }
};

struct AlignCtx {
    lm_index *ix;
    lm_qbatch *qb;
    Work *w;
    lm_stage_stats *stats;
    const uint8_t *wb = nullptr; // window buffer biased so that wb + task.woff addresses this chunk's windows
    int64_t wfa_budget = (int64_t)40 << 30;

    // per chunk device buffers
    DBuf<int32_t> wlen;
    DBuf<int64_t> woff;
    DBuf<uint8_t> wbuf;
    // windows of the tasks with chains, compact, gathered over the chunks of one extendMatch / WFA round
    DBuf<uint8_t> gwbuf; // the round's compact window buffer
    DBuf<int32_t> gw_idx;
    DBuf<int64_t> gw_dest;
    DBuf<unsigned long long> pa_count;
    int64_t pa_cap = 0; // running estimate of the anchors per chunk
    double pa_ratio_own = 0, *pa_ratio = &pa_ratio_own; // anchors per window byte seen so far (shared by the two contexts)
    DBuf<int64_t> pa_off;
    DBuf<uint64_t> A0, B0, A1, B1;
    DBuf<LmSub> subs;
    DBuf<uint8_t> marks;
    DBuf<uint64_t> msi;
    DBuf<int32_t> stack, out_n, clr_n;
    DBuf<LmChain2> out, out_compact;
    DBuf<int64_t> res_off;
    DBuf<Task> tasks;
    DBuf<HspIn> hsp_in;
    DBuf<HspExt> hsp_ext;
    DBuf<int32_t> ext_cap, ext_wcap, ext_msi;
    DBuf<int64_t> ext_off;
    DBuf<uint16_t> ext_subs;
    DBuf<uint64_t> ext_rows;   // grid chainer: per resident wavefront [LM_EXT_ROWS][64] x 128-bit row masks
    DBuf<uint32_t> ext_rstart; // first anchor index of each row
    DBuf<WfaIn> wfa_in;
    DBuf<WfaOut> wfa_out;
    DBuf<uint64_t> ops_pool;
    // the LDS WFA passes of the four length classes run side by side (a pass ends in a tail of a few long alignments that
    // leaves most CUs idle): each class has its own stream, queue and scratch
    struct LeanCtx {
        hipStream_t st = nullptr;
        DBuf<int32_t> todo, hdr_pool, arena_pool;
        DBuf<unsigned int> queue;
        DBuf<uint8_t> tmp;
        ~LeanCtx() {
            if (st) (void)hipStreamDestroy(st);
        }
    } lean[2 * LM_WFA_CLASSES]; // per length class: the chain that starts at the class's ring width, and the one of the problems predicted wider
    // the global-memory WFA fallback runs beside the LDS passes of the shorter length classes: own stream and buffers
    struct WideCtx {
        hipStream_t st = nullptr;
        DBuf<WfaIn> in;
        DBuf<WfaOut> out;
        DBuf<int32_t> todo, hdr, arena;
        DBuf<uint64_t> ops;
        DBuf<uint8_t> tmp;
        ~WideCtx() {
            if (st) (void)hipStreamDestroy(st);
        }
    } wide;
    AlignCtx() {
        for_each_phase([](auto &b) { b.phase = true; });
    }
    ~AlignCtx() {
    }
    template <class F> void for_each_phase(F f) {
        f(wlen); f(woff); f(wbuf); f(gwbuf); f(gw_idx); f(gw_dest); f(pa_off); f(A0); f(B0); f(A1); f(B1); f(subs);
        f(marks); f(msi); f(stack); f(out_n); f(clr_n); f(out); f(out_compact); f(res_off); f(tasks); f(hsp_in);
        f(hsp_ext); f(ext_cap); f(ext_wcap); f(ext_msi); f(ext_off); f(ext_subs); f(ext_rows); f(ext_rstart);
        f(wfa_in); f(wfa_out);  f(ops_pool);
        f(wide.in); f(wide.out); f(wide.todo); f(wide.hdr); f(wide.arena); f(wide.ops); f(wide.tmp);
        for (auto &l : lean) { f(l.todo); f(l.hdr_pool); f(l.arena_pool); f(l.tmp); }
    }
    // the alignment half is over: its buffers go back to the handle's scratch arena (the seeding half of the next batch
    // part is carved from the same slabs; both halves sized to their shares of the scratch budget do not fit side by side)
    int64_t release_big(int64_t keep_below_bytes) {
        int64_t freed = 0;
        for_each_phase([&](auto &b) {
            if (b.arena || (int64_t)b.bytes() > keep_below_bytes) {
                freed += (int64_t)b.bytes();
                b.release();
            }
        });
        return freed;
    }
};

} // namespace lm
void lm_free_align_ctx(lm_index *ix, int lane) { // lane < 0: both
    if (lane != 1)
        for (auto &c : ix->actx) {
            delete c;
            c = nullptr;
        }
    if (lane != 0)
        for (auto &c : ix->actx1) {
            delete c;
            c = nullptr;
        }
}
namespace lm {

struct HspMeta { // host-side view of one WFA problem
    float est_div; // divergence implied by the pseudo-alignment identity (scratch sizing only)
    int64_t task;
    uint32_t q;
    HspIn in;
    HspExt ext;
};

// Runs pseudo-alignment for tasks[t0,t1) (host copy `ht`), returns per task the Chain2 results.
// `ht`: host copy of the tasks; `dev_tasks` their device copy (null: upload `ht`); `base` = window offset of the first
// task (the tasks carry offsets into one virtual buffer of all windows of the batch, a chunk uses a slice of it)
// `own_windows`: the targets are caller-provided ASCII windows in a.wbuf (stage-level entry point). The search path passes
// false: the anchor kernel takes its k-mers from the 2-bit genomes, and the ASCII windows of the tasks that produce chains
// are extracted later, compactly, by the consumer (align_range).
static void run_pseudo(AlignCtx &a, TaskSpan ht, std::vector<int64_t> &res_off_h, std::vector<LmChain2> &res_h,
                       const Task *dev_tasks = nullptr, int64_t base = 0, bool own_windows = true) {
    lm_index *ix = a.ix;
    lm_qbatch *qb = a.qb;
    int64_t nt = (int64_t)ht.size();
    res_off_h.assign(nt + 1, 0);
    res_h.clear();
    if (nt == 0) return;
    int64_t W = ht.back().woff + ht.back().wlen - base;
    if (own_windows) {
        a.wbuf.ensure((size_t)W + 64);
        a.wb = a.wbuf.p - base;
    } else {
        a.wb = nullptr;
    }
    const Task *tasks_d = dev_tasks;
    if (!tasks_d) {
        a.tasks.ensure((size_t)nt);
        HIPCHK(hipMemcpyAsync(a.tasks.p, ht.p, sizeof(Task) * nt, hipMemcpyHostToDevice, S(ix)));
        tasks_d = a.tasks.p;
    }
    a.stats->window_bases += W;
    // key layout: when task number + anchor fields fit 64 bits the anchors are single compact keys (keys-only sort)
    int abits = 1, qbits = 1, tbits = 1;
    {
        while (((int64_t)1 << abits) < nt + 1) abits++;
        int maxw = 1, maxq = 1;
        for (int64_t i = 0; i < nt; i++) maxw = std::max(maxw, ht[i].wlen);
        for (int q = 0; q < qb->nq; q++) maxq = std::max<int>(maxq, (int)(qb->h_qoff[q + 1] - qb->h_qoff[q]));
        while ((1 << tbits) <= maxw + 64) tbits++; // reverse-strand anchors start up to K bases past the last k-mer
        while ((1 << qbits) <= maxq) qbits++;
    }
    const bool compact = abits + qbits + 6 + tbits + 2 <= 64 && !getenv("LM_DEBUG_PA_WIDE_KEYS"); // 2nd: test hook
    const int key_bits = abits + qbits + 6 + tbits + 2, sh_a = qbits + 6 + tbits + 2;
    // single pass: anchors appended to (A0, B0) in arbitrary order; the buffer size is a running estimate, the kernel
    // counts past it, so an undersized buffer costs one re-run
    a.pa_count.ensure(2 + LM_PA_MAX_SEGS); // anchors, the filter's group counter, candidates per segment
    if (a.pa_cap < (int64_t)1 << 20) a.pa_cap = std::max<int64_t>((int64_t)1 << 20, W / 8);
    if (const char *e = getenv("LM_DEBUG_PA_CAP")) a.pa_cap = std::max<int64_t>(1, atoll(e)); // test hook: force the re-run
    int64_t TP = 0;
    int nseg = 1;
    // candidate segments by task-group range + XCD-local search pay when a query's comparison tables (12 B per k-mer and
    // strand + the bucket table) are a sizeable part of an XCD's 4-MB L2: long reads.  For gene-sized queries (tables of tens of
    // KB: they stay in L2 anyway) the many small blocks per range only cost (C2: k_pa_search 4.2 -> 6.4 ms), so those keep
    // the plain layout
    bool by_group = ix->tune.pa_seg_by_group != 0;
    {
        int64_t maxq = 0;
        for (int q = 0; q < qb->nq; q++) maxq = std::max<int64_t>(maxq, qb->h_qoff[q + 1] - qb->h_qoff[q]);
        if (maxq < 8192) by_group = false;
    }
    bool nseg_stale = false; // the layout changed (task-group ranges -> per wavefront): count the segments again
    for (int attempt = 0;; attempt++) {
        if (!compact) a.A0.ensure((size_t)a.pa_cap);
        a.B0.ensure((size_t)a.pa_cap);
        a.B1.ensure((size_t)a.pa_cap); // the sort's second buffer holds the candidate list of k_pa_filter until then
        // the candidate list in segments with a counter each (k_pa_filter), at least 4096 entries per segment; segments by
        // task group: never more segments than groups, and a re-run after an overflow keeps the number of segments (the
        // measured fullest segment then sizes the next attempt exactly)
        if (!by_group) {
            // (once per layout: a re-run after an overflow keeps the number of segments, so that the capacity of a segment grows
            // with the buffer.  Recomputed every attempt it stayed at ~4096 entries while the buffer grew - a chunk whose busiest
            // wavefront found a few more than that overflowed five times in a row and the search failed, depending on which
            // wavefront happened to take which slices: round 6, seen once in three runs of the test suite)
            if (attempt == 0 || nseg_stale) nseg = (int)std::max<int64_t>(1, std::min<int64_t>(1024, a.pa_cap / 4096));
            nseg_stale = false;
        } else if (attempt == 0) { // ranges of task groups x LM_PA_RANGE_SEGS segments each
            const int64_t ngroups = (nt + LM_PA_GROUP - 1) / LM_PA_GROUP;
            const int64_t ranges = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(LM_PA_MAX_SEGS / LM_PA_RANGE_SEGS, ngroups),
                                                                          a.pa_cap / 4096 / LM_PA_RANGE_SEGS));
            nseg = (int)ranges * LM_PA_RANGE_SEGS;
        }
        const int64_t seg_cap = a.pa_cap / nseg;
        HIPCHK(hipMemsetAsync(a.pa_count.p, 0, (size_t)(2 + nseg) * sizeof(unsigned long long), S(ix)));
        {
            Prof p(ix, "k_pa_filter", W);
            launch_pa_filter(S(ix), ix->view, tasks_d, nt, a.wb, qb->d_posoff.p, a.w->nvalid.p, a.w->cmp_bits.p,
                             qb->d_bits_off.p, qb->d_bits_log.p, ix->host.k, 11, a.pa_count.p + 2, nseg, seg_cap, a.B1.p,
                             a.pa_count.p + 1, device_cus(ix->device), by_group ? 1 : 0, ix->tune.pa_filter_roll != 0);
        }
        {
            Prof p(ix, "k_pa_search");
            launch_pa_search(S(ix), ix->view, tasks_d, a.wb, a.w->k_cmp, a.w->v_cmp, qb->d_posoff.p, a.w->nvalid.p,
                             a.w->cmp_tab.p, qb->d_tab_off.p, qb->d_tab_bits.p, ix->host.k, 11, a.pa_count.p + 2, nseg, seg_cap,
                             a.B1.p, a.pa_count.p, a.pa_cap, a.A0.p, a.B0.p, compact ? qbits : 0, compact ? tbits : 0,
                             by_group ? 1 : 0);
        }
        std::vector<unsigned long long> hv((size_t)2 + nseg);
        HIPCHK(hipMemcpyAsync(hv.data(), a.pa_count.p, hv.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, S(ix)));
        sync(ix);
        TP = (int64_t)hv[0];
        int64_t ncand = 0, seg_max = 0;
        for (int j = 0; j < nseg; j++) {
            ncand += (int64_t)hv[2 + j];
            seg_max = std::max<int64_t>(seg_max, (int64_t)hv[2 + j]);
        }
        {   // algorithmic bytes of the search (align stage): 8 B per candidate read, 8 B per anchor written, and
            // the comparison index of every query of the chunk once (12 B per indexed k-mer, both strands)
            int64_t idx = 0;
            uint32_t prevq = 0xffffffffu;
            for (int64_t i = 0; i < nt; i++)
                if (ht[i].q != prevq) { // tasks are in (query, genome) order
                    prevq = ht[i].q;
                    idx += 24 * std::max<int64_t>(0, qb->h_qoff[prevq + 1] - qb->h_qoff[prevq] - (ix->host.k - 1));
                }
            prof_add_bytes(ix, "k_pa_search", 8 * std::min<int64_t>(ncand, a.pa_cap) + 8 * std::min<int64_t>(TP, a.pa_cap) + idx);
        }
        dbg_stamp("pseudo-alignment anchors of a chunk done");
        if (getenv("LM_DEBUG"))
            fprintf(stderr, "[lm] pseudo-alignment: %lld window bases, %lld candidates (fullest of %d segments: %lld of %lld), %lld anchors\n",
                    (long long)W, (long long)ncand, nseg, (long long)seg_max, (long long)seg_cap, (long long)TP);
        // candidates and anchors share the estimate; a segment that overflowed dropped candidates: size for the fullest
        int64_t need = std::max<int64_t>(seg_max > seg_cap ? seg_max * nseg + seg_max * nseg / 8 : ncand, TP);
        if (need <= a.pa_cap && seg_max <= seg_cap) break;
        if (attempt > 4)
            throw HipError("pseudo-alignment anchor buffer keeps overflowing (windows " + std::to_string(nt) + ", window bases " + std::to_string(W) +
                           ", candidates " + std::to_string(ncand) + ", fullest of " + std::to_string(nseg) + " segments " + std::to_string(seg_max) + " of " +
                           std::to_string(seg_cap) + ", anchors " + std::to_string(TP) + ", capacity " + std::to_string(a.pa_cap) + ")");
        if (by_group && seg_max > seg_cap && seg_max * nseg > 3 * std::max<int64_t>(std::max(ncand, TP), 1)) {
            // Segments by task-group range are as uneven as the batch: one 200-kb plasmid query among genes (or the reads of a
            // mixed batch) fills its range with 10-50 x the mean, and a uniform segment capacity sized for the fullest range
            // would blow the whole list - and the anchor buffers that share its size - up by that factor (an out-of-memory
            // or a halved chunk exactly on the mixed workloads).  Such a chunk takes the balanced per-wavefront layout.
            by_group = false;
            nseg_stale = true;
            need = std::max<int64_t>(ncand + ncand / 8, TP);
            if (getenv("LM_DEBUG")) fprintf(stderr, "[lm] pseudo-alignment: uneven ranges (fullest %lld x %d segments vs %lld candidates): per-wavefront segments for this chunk\n",
                                            (long long)seg_max, nseg, (long long)ncand);
        }
        a.pa_cap = std::max<int64_t>(need + need / 8, a.pa_cap + a.pa_cap / 4);
    }
    // anchors per window byte of what has been seen (sizes the next chunks so that they need no halving); small chunks
    // (the odd tasks left behind a halving) say little
    if (W > ((int64_t)64 << 20)) *a.pa_ratio = std::max(*a.pa_ratio * 0.9, (double)TP / (double)W);
    else if (W > 0 && *a.pa_ratio == 0) *a.pa_ratio = (double)TP / (double)W;
    if (TP >= (int64_t)1 << 31 || (BUDGET(ix) > 0 && TP * 90 > BUDGET(ix) * 13 / 100)) {
        if (nt <= 1) throw HipError("too many pseudo-alignment anchors for one chain window");
        a.stats->window_bases -= W; // the chunk comes back in halves
        throw ChunkTooLarge();
    }
    a.stats->pa_anchors += TP;
    a.pa_off.ensure((size_t)nt + 2);
    a.out_n.ensure((size_t)nt + 1);
    a.clr_n.ensure((size_t)nt + 1);
    HIPCHK(hipMemsetAsync(a.out_n.p, 0, sizeof(int32_t) * (nt + 1), S(ix)));
    if (TP > 0) {
        if (!compact) a.A1.ensure((size_t)a.pa_cap);
        a.B1.ensure((size_t)a.pa_cap);
        if (compact) {
            Prof p(ix, "sort_pa_anchors", TP * 16); // one u64 key in and out
            prim_sort_keys(S(ix), TMP(ix), a.B0.p, a.B1.p, (size_t)TP, 0, key_bits);
            std::swap(a.B0.p, a.B1.p); // sorted keys are what follows calls B0
            std::swap(a.B0.cap, a.B1.cap);
            launch_pa_task_off_sorted(S(ix), a.B0.p, sh_a, TP, nt, a.pa_off.p);
        } else {
            Prof p(ix, "sort_pa_anchors", TP * 32);
            sort_anchors_fields(ix, a.A0.p, a.B0.p, a.A1.p, a.B1.p, TP, abits, qbits, tbits);
            std::swap(a.A0.p, a.A1.p); // the sorted list is in (A1, B1): make it (A0, B0) for what follows
            std::swap(a.A0.cap, a.A1.cap);
            std::swap(a.B0.p, a.B1.p);
            std::swap(a.B0.cap, a.B1.cap);
            launch_pa_task_off_sorted(S(ix), a.A0.p, 0, TP, nt, a.pa_off.p);
        }
        a.subs.ensure((size_t)TP);
        a.marks.ensure((size_t)TP);
        a.msi.ensure((size_t)TP);
        a.stack.ensure(2 * (size_t)TP + 5 * (size_t)nt + 48); // (+ the list of long windows, its counter, the debug counters)
        a.out.ensure((size_t)TP);
        LmChain2Opt o2; // search.go:364-378
        o2.max_gap = ix->opt.align_max_gap;
        o2.min_score = (int)((double)ix->opt.align_min_match_len * ix->opt.align_min_pident / 100);
        o2.min_align_len = ix->opt.align_min_match_len;
        o2.band_base = ix->opt.align_band;
        o2.band_count = ix->opt.align_band / 2;
        o2.heuristic_pident = 15;
        {
            Prof p(ix, "k_pa_chain", TP * 32);
            launch_pa_chain(S(ix), a.B0.p, a.pa_off.p, nt, ix->host.k, o2, a.subs.p, a.marks.p, a.msi.p, a.stack.p,
                            a.out.p, a.out_n.p, a.clr_n.p, compact ? qbits : 0, compact ? tbits : 0);
        }
        a.res_off.ensure((size_t)nt + 2);
        int64_t NR = scan_to_i64<int32_t, CastI32>(ix, a.out_n.p, nt, a.res_off.p);
        if (NR > 0) {
            a.out_compact.ensure((size_t)NR);
            launch_gather_chain2(S(ix), a.out.p, a.pa_off.p, a.out_n.p, a.res_off.p, nt, a.out_compact.p);
            d2h(ix, res_h, a.out_compact.p, (size_t)NR);
        }
        d2h(ix, res_off_h, a.res_off.p, (size_t)nt + 1);
        sync(ix);
    }
}

// algorithmic bytes of WFA: both sequences are read once, 2*(len_q+len_t)
static int64_t wfa_bytes(const std::vector<WfaIn> &in, const std::vector<int32_t> &ids) {
    int64_t b = 0;
    for (int32_t i : ids) b += 2ll * ((int64_t)in[i].qlen + in[i].tlen);
    return b;
}

// WFA for a list of problems already described by device pointers; retries with more memory on overflow.
// expected number of wavefront cells (M+I+D) up to score s: the width grows by 2 every gap-open step until the
// wf-adaptive cut-off (max distance 50) caps it at ~110 diagonals
static int64_t wfa_cells(int64_t s) {
    if (s <= 444) return 3 * (s + s * s / 8 + 1);
    return 3 * (444 + 24642 + (s - 444) * 112);
}
// divergence implied by a pseudo-alignment identity (fraction of bases covered by exact >=11-mers):
// f(d) = (1-d)^11 (1+11d), inverted by bisection
static double div_from_pseudo_pident_slow(double pid) {
    double x = pid / 100.0, lo = 0.0, hi = 0.6;
    for (int it = 0; it < 40; it++) {
        double mid = 0.5 * (lo + hi);
        double f = std::pow(1.0 - mid, 11) * (1.0 + 11.0 * mid);
        if (f > x)
            lo = mid;
        else
            hi = mid;
    }
    return hi;
}

static double div_from_pseudo_pident(double pid) { // table over integer percent, built once
    static float lut[102];
    static bool init = false;
    if (!init) {
        for (int i = 0; i <= 101; i++) lut[i] = (float)div_from_pseudo_pident_slow((double)i);
        init = true;
    }
    int i = (int)pid;
    if (i < 0) i = 0;
    if (i > 100) i = 100;
    return lut[i]; // floor of the identity => slightly over-estimated divergence
}

static void run_wfa(AlignCtx &a, std::vector<WfaIn> &in, std::vector<WfaOut> &out, std::vector<uint64_t> &ops_h,
                    std::vector<int64_t> &ops_off_h, bool want_ops, const std::vector<float> *est_div = nullptr) {
    lm_index *ix = a.ix;
    const int lane = tls_lane;
    int64_t n = (int64_t)in.size();
    out.assign(n, WfaOut());
    ops_off_h.assign(n + 1, 0);
    ops_h.clear();
    if (n == 0) return;
    // scratch: the LDS passes and the global-memory fallback run at the same time
    const int64_t lean_budget = BUDGET(ix) > 0 ? std::min<int64_t>(a.wfa_budget, BUDGET(ix) * 26 / 100) : a.wfa_budget;
    const int64_t wide_budget = BUDGET(ix) > 0 ? std::min<int64_t>((int64_t)72 << 30, BUDGET(ix) * 10 / 100) : (int64_t)72 << 30;
    a.wfa_out.ensure((size_t)n);
    a.wfa_in.ensure((size_t)n);
    std::vector<std::vector<uint64_t>> ops_keep(want_ops ? n : 0);
    std::atomic<int64_t> retries{0};

    // ---- global-memory fallback (k_wfa_wave: one wavefront per problem, ring and wavefronts in global memory) for what the
    // LDS kernels cannot hold: wavefronts wider than 510 diagonals, sequences above 65 kb or with non-ACGT bytes, scratch
    // overflow.  Works on its own copies (problem descriptors, output, scratch) on the calling thread's stream.
    auto wide_run = [&](std::vector<int32_t> todo, std::vector<int32_t> level0, AlignCtx::WideCtx &wc) {
        std::vector<WfaIn> in2(in);
        std::vector<int32_t> level(n, 0);
        for (size_t j = 0; j < todo.size(); j++) level[todo[j]] = level0[j];
        wc.in.ensure((size_t)n);
        wc.out.ensure((size_t)n);
        while (!todo.empty()) {
            // take a prefix of todo that fits the scratch budget
            std::vector<int32_t> cur;
            int64_t hdr_tot = 0, arena_tot = 0, ops_tot = 0;
            size_t taken = 0;
            for (; taken < todo.size(); taken++) {
                int32_t i = todo[taken];
                WfaIn &w = in2[i];
                int64_t L = (int64_t)w.qlen + w.tlen;
                // first guess from the divergence estimate (pseudo-alignment identity) with 40% head-room, every retry
                // doubles the score bound (the cell estimate follows it)
                double dv = est_div ? (double)(*est_div)[i] : 0.12;
                int64_t ms = (int64_t)(96 + 1.4 * dv * 4.6 * (double)(L / 2) + 0.05 * (double)L) << level[i];
                if (ms > 8 * L + 64) ms = 8 * L + 64; // a global alignment never exceeds this penalty
                int64_t ar = std::max<int64_t>(4096, wfa_cells(ms) + wfa_cells(ms) / 4) << (level[i] > 2 ? level[i] - 2 : 0);
                int64_t oc = std::min<int64_t>(L + 2, (int64_t)(128 + 3.0 * dv * (double)L) << level[i]);
                int64_t need = (ms * 9 + ar) * 4 + oc * 8;
                if (!cur.empty() && (hdr_tot * 9 + arena_tot) * 4 + ops_tot * 8 + need > wide_budget) break;
                w.max_score = (int32_t)std::min<int64_t>(ms, 2000000000);
                w.hdr_off = hdr_tot * 9;
                w.arena_off = arena_tot;
                w.arena_cap = ar;
                w.ops_off = ops_tot;
                w.ops_cap = (int32_t)oc;
                hdr_tot += ms;
                arena_tot += ar;
                ops_tot += oc;
                cur.push_back(i);
            }
            todo.erase(todo.begin(), todo.begin() + taken);
            wc.hdr.ensure((size_t)hdr_tot * 9 + 16);
            wc.arena.ensure((size_t)arena_tot + 16);
            wc.ops.ensure((size_t)ops_tot + 16);
            wc.todo.ensure(cur.size());
            HIPCHK(hipMemcpyAsync(wc.in.p, in2.data(), sizeof(WfaIn) * n, hipMemcpyHostToDevice, S(ix)));
            HIPCHK(hipMemcpyAsync(wc.todo.p, cur.data(), sizeof(int32_t) * cur.size(), hipMemcpyHostToDevice, S(ix)));
            {
                Prof p(ix, "k_wfa_wide", wfa_bytes(in2, cur));
                launch_wfa_wide(S(ix), wc.in.p, n, wc.todo.p, (int64_t)cur.size(), wc.hdr.p, wc.arena.p, wc.ops.p, wc.out.p);
            }
            std::vector<WfaOut> tmp;
            d2h(ix, tmp, wc.out.p, (size_t)n);
            std::vector<uint64_t> ops_tmp;
            if (want_ops) d2h(ix, ops_tmp, wc.ops.p, (size_t)ops_tot);
            sync(ix);
            for (int32_t i : cur) {
                if (tmp[i].r.status == 1 || tmp[i].r.status == 3) {
                    level[i]++;
                    retries++;
                    if (level[i] > 12) throw HipError("WFA scratch overflow after 12 retries");
                    todo.push_back(i);
                } else {
                    out[i] = tmp[i];
                    if (want_ops && tmp[i].r.nops > 0)
                        ops_keep[i].assign(ops_tmp.begin() + in2[i].ops_off, ops_tmp.begin() + in2[i].ops_off + tmp[i].r.nops);
                }
            }
        }
    };

    // problems in decreasing order of expected cost (divergence estimate x length): bucket sort
    std::vector<int32_t> order(n);
    int64_t ops_tot = 0;
    {
        const int NB = 1024;
        std::vector<float> key(n);
        float kmax = 1e-9f;
        for (int64_t i = 0; i < n; i++) {
            int64_t L = (int64_t)in[i].qlen + in[i].tlen;
            float dv = est_div ? (*est_div)[i] : 0.12f;
            key[i] = (dv + 0.01f) * (float)L;
            kmax = std::max(kmax, key[i]);
        }
        std::vector<int32_t> cnt(NB + 1, 0);
        std::vector<int16_t> bk(n);
        for (int64_t i = 0; i < n; i++) {
            int b = NB - 1 - (int)(key[i] / kmax * (NB - 1)); // bucket 0 = most expensive
            bk[i] = (int16_t)b;
            cnt[b + 1]++;
        }
        for (int b = 0; b < NB; b++) cnt[b + 1] += cnt[b];
        for (int64_t i = 0; i < n; i++) order[cnt[bk[i]]++] = (int32_t)i;
    }
    for (int64_t i = 0; i < n; i++) {
        WfaIn &w = in[i];
        int64_t L = (int64_t)w.qlen + w.tlen;
        double dv = est_div ? (double)(*est_div)[i] : 0.12;
        int64_t oc = std::min<int64_t>(L + 2, (int64_t)(128 + 3.0 * dv * (double)L));
        w.hdr_off = w.arena_off = w.arena_cap = 0;
        w.max_score = 0;
        w.ops_off = ops_tot;
        w.ops_cap = (int32_t)oc;
        ops_tot += oc;
    }
    a.ops_pool.ensure((size_t)ops_tot + 16);
    HIPCHK(hipMemcpyAsync(a.wfa_in.p, in.data(), sizeof(WfaIn) * n, hipMemcpyHostToDevice, S(ix)));
    sync(ix); // the class streams read the descriptors
    std::vector<int32_t> fb_items, fb_level; // what leaves the LDS kernels, with its starting scratch level
    std::mutex fb_mu;
    // length classes (sequence words of 16 bases): they pick the ring width a problem starts with, whether the kernel keeps
    // the whole packed sequences in LDS or reads them through sliding windows (lm_tune::wfa_win), and keep the few long
    // alignments of a round off the queue of the many short ones.  The last class (beyond 65 kb: what the whole-sequence
    // kernel cannot hold) is open-ended and always windowed.  Within a class the queue keeps the longest-expected-first order
    // (Measured in round 4 and removed in round 6: a second chain of passes per class for the problems |tlen - qlen| predicts to
    // outgrow the class's ring, started at the predicted width - the retry passes leave the critical chain, but the round is
    // bound by the sum of the work: C3 22.6 s against 13.0 s per step.)
    constexpr int NCLS = LM_WFA_CLASSES, NCH = LM_WFA_CLASSES;
    const int bounds[NCLS - 1] = {128, 512, 2048, 4096};
    std::vector<int32_t> cls[NCH];
    int cw[NCH];
    int64_t cl[NCH], cs[NCH];
    for (int c = 0; c < NCH; c++) {
        cw[c] = 1;
        cl[c] = 1;
        cs[c] = 0;
    }
    int first_nc[NCH];
    bool win[NCH];
    for (int c = 0; c < NCLS; c++) {
        first_nc[c] = ix->tune.wfa_first_nc[c];
        win[c] = ix->tune.wfa_win[c] != 0 || c == NCLS - 1;
    }
    for (int32_t i : order) {
        const int wds = (std::max(in[i].qlen, in[i].tlen) + 15) / 16;
        int c = 0;
        while (c < NCLS - 1 && wds > bounds[c]) c++;
        cls[c].push_back(i);
        cw[c] = std::max(cw[c], wds);
        const int64_t L = (int64_t)in[i].qlen + in[i].tlen;
        cl[c] = std::max<int64_t>(cl[c], L);
        // the score this problem is expected to reach: ~2.5 x divergence x (len_q + len_t) at the 4/6/2 penalties, with a
        // factor two on top (what goes beyond reports a scratch overflow and is aligned by the fallback)
        const double dv = est_div ? (double)(*est_div)[i] : 0.12;
        cs[c] = std::max<int64_t>(cs[c], (int64_t)(5.0 * (dv + 0.01) * (double)L) + 2048);
    }
    // scratch of the classes side by side: what each would like (resident wavefronts x expected backtrace bytes of its
    // longest problem), scaled down together when that exceeds the lean share of the budget
    int64_t want[NCH], share[NCH];
    int64_t want_tot = 0;
    for (int c = 0; c < NCH; c++) {
        const int64_t m = (int64_t)cls[c].size();
        const int64_t smax = std::min<int64_t>(8 * cl[c] + 64, cs[c]);
        const int64_t per = (smax / 2 + 2) * 64 * first_nc[c] + 2 * cl[c] + 4096 + (smax / 2 + 4) * 16;
        want[c] = m == 0 ? 0 : std::min<int64_t>(m, wfa_resident_blocks(ix->device, cw[c], first_nc[c], win[c], ix->tune.wfa_r16 && wfa_r16_ok(cw[c], first_nc[c], win[c]))) * per * 9 / 8;
        want_tot += want[c];
    }
    for (int c = 0; c < NCH; c++)
        share[c] = want_tot <= lean_budget ? std::max<int64_t>(want[c], (int64_t)64 << 20)
                                           : std::max<int64_t>((int64_t)((double)want[c] / (double)want_tot * (double)lean_budget), (int64_t)64 << 20);
    // one launch of the persistent LDS kernel per length class and ring width.  Resident wavefronts per CU: windowed, set by
    // the ring width alone (128 diagonals: 23, 256: 14, 512: 7, 1024: 4); whole sequences in LDS, by the longest problem of
    // the launch too (gene-sized HSPs at 128 diagonals: 28, 8-32 kb at 256: 6)
    auto persistent_pass = [&](AlignCtx::LeanCtx &lc, int64_t budget, const std::vector<int32_t> &items, int seq_words, bool use_win,
                               int64_t lmax, int64_t s_expect, std::vector<int32_t> &too_wide, int nc) {
        const int64_t m = (int64_t)items.size();
        if (m == 0) return;
        const bool r16 = ix->tune.wfa_r16 && wfa_r16_ok(seq_words, nc, use_win); // 16-bit ring cells: more wavefronts per CU
        const int resident = wfa_resident_blocks(ix->device, seq_words, nc, use_win, r16);
        int nblocks = (int)std::min<int64_t>(m, std::max<int64_t>(256, (int64_t)resident * ix->tune.wfa_resident_pct / 100));
        // private scratch per resident wave: one backtrace byte per wavefront cell + 8 bytes per even score; never
        // more than the longest problem of the class is expected to need
        const int64_t smax = std::min<int64_t>(8 * lmax + 64, s_expect); // a global alignment never exceeds 8 per base
        int64_t bytes = budget / nblocks * 7 / 8;
        bytes = std::min<int64_t>(bytes, (smax / 2 + 2) * 64 * nc + 2 * lmax + 4096);
        bytes = std::max<int64_t>(bytes, 65536);
        bytes = std::min<int64_t>(bytes, 2000000000) & ~(int64_t)15;
        int64_t entries = std::min<int64_t>(smax / 2 + 4, bytes / 24 + 1024);
        if (getenv("LM_DEBUG"))
            fprintf(stderr, "[lm +%.1f ms] wfa pass (%d diagonals, %s) problems=%lld blocks=%d (resident %d) bytes/block=%lld scores=%lld longest=%lld\n",
                    now_ms() - g_dbg_t0, 64 * nc, use_win ? "windows" : "whole sequences", (long long)m, nblocks, resident, (long long)bytes,
                    (long long)(2 * entries), (long long)lmax);
        lc.hdr_pool.ensure((size_t)(entries * 2) * nblocks + 16);
        lc.arena_pool.ensure((size_t)(bytes / 4) * nblocks + 16);
        lc.todo.ensure((size_t)m);
        lc.queue.ensure(1);
        HIPCHK(hipMemcpyAsync(lc.todo.p, items.data(), sizeof(int32_t) * m, hipMemcpyHostToDevice, S(ix)));
        HIPCHK(hipMemsetAsync(lc.queue.p, 0, sizeof(unsigned int), S(ix)));
        {
            static const char *const names[2][5] = {{"k_wfa_lean64", "k_wfa_lean", "k_wfa_lean256", "k_wfa_lean512", "k_wfa_lean1024"},
                                                    {"k_wfa_win64", "k_wfa_win128", "k_wfa_win256", "k_wfa_win512", "k_wfa_win1024"}};
            Prof p(ix, names[use_win ? 1 : 0][nc == 16 ? 4 : nc == 8 ? 3 : nc == 4 ? 2 : nc == 1 ? 0 : 1], wfa_bytes(in, items));
            launch_wfa(S(ix), a.wfa_in.p, n, lc.todo.p, m, nblocks, lc.hdr_pool.p, entries * 2, (uint8_t *)lc.arena_pool.p, bytes,
                       a.ops_pool.p, lc.queue.p, seq_words, want_ops ? 1 : 0, a.wfa_out.p, nc, use_win, r16, nullptr);
        }
        sync(ix);
        // this pass's results: the records of its items (other classes write theirs into the same array meanwhile)
        std::vector<WfaOut> tmp;
        d2h(ix, tmp, a.wfa_out.p, (size_t)n);
        std::vector<uint64_t> ops_tmp;
        if (want_ops) d2h(ix, ops_tmp, a.ops_pool.p, (size_t)ops_tot);
        sync(ix);
        int64_t n3 = 0, n1 = 0;
        for (int32_t i : items) {
            int stt = tmp[i].r.status;
            n3 += stt == 3;
            n1 += stt == 1;
            if (stt == 3 && tmp[i].r.score == 0) { // not plain ACGT (a width overflow reports the width): no ring width helps,
                std::lock_guard<std::mutex> l(fb_mu); // the byte-comparing kernel takes it at once
                fb_items.push_back(i);
                fb_level.push_back(1);
            } else if (stt == 3) { // wider than this ring
                too_wide.push_back(i);
            } else if (stt == 1) { // scratch or ops overflow: per-problem scratch in the global-memory kernel
                std::lock_guard<std::mutex> l(fb_mu);
                fb_items.push_back(i);
                fb_level.push_back(1);
                retries++;
            } else {
                out[i] = tmp[i];
                if (want_ops && tmp[i].r.nops > 0)
                    ops_keep[i].assign(ops_tmp.begin() + in[i].ops_off, ops_tmp.begin() + in[i].ops_off + tmp[i].r.nops);
            }
        }
        if (getenv("LM_DEBUG"))
            fprintf(stderr, "[lm +%.1f ms] wfa pass (%d diagonals, %lld problems) done: %lld wider than %d diagonals (or non-ACGT), %lld on scratch overflow\n",
                    now_ms() - g_dbg_t0, 64 * nc, (long long)m, (long long)n3, 64 * nc - 2, (long long)n1);
    };
    // ring width by experience: alignments of tens of kb at ONT error rates run wavefronts of several hundred diagonals
    // under wf-adaptive(10,50) (all of the >= 32-kb class and two thirds of the 8-32-kb class outgrow 126), gene-sized ones
    // stay below 126.  A pass that turns out too narrow returns status 3 and the next width takes over.  (The width a
    // problem ends up needing is |tlen - qlen| + a few dozen diagonals - the wavefront must reach the final diagonal; of
    // 78 000 c3-shaped problems exactly the ones with |tlen - qlen| >= 210..260 outgrew 254 diagonals, >= 505 outgrew 510 -
    // but STARTING there was measured slower, 2.48 s vs 1.99 s per step: a failed narrow attempt costs little, the wide
    // rings are slower per score and their LDS keeps the anchor filter's workgroups off the CUs.)
    auto class_chain = [&](int c) {
        AlignCtx::LeanCtx &lc = a.lean[c];
        std::vector<int32_t> cur = cls[c], next;
        for (int nc = first_nc[c]; nc <= 16 && !cur.empty(); nc *= 2) { // what a pass leaves (status 3) goes to the next width
            next.clear();
            persistent_pass(lc, share[c], cur, cw[c], win[c], cl[c], cs[c], next, nc);
            cur.swap(next);
        }
        std::lock_guard<std::mutex> l(fb_mu);
        for (int32_t i : cur) { // the hard ones: generous scratch at once instead of an overflow and a second launch
            fb_items.push_back(i);
            fb_level.push_back(2);
            retries++;
        }
    };
    std::thread wide_thread;
    std::exception_ptr wide_err;
    auto start_wide = [&]() { // what has left the LDS kernels so far goes to the fallback on its own stream and thread
        std::vector<int32_t> items, level;
        {
            std::lock_guard<std::mutex> l(fb_mu);
            if (fb_items.empty() || wide_thread.joinable()) return;
            items.swap(fb_items);
            level.swap(fb_level);
        }
        wide_thread = std::thread([&, items, level]() {
            try {
                HIPCHK(hipSetDevice(ix->device));
                tls_lane = lane;
                tls_stream = a.wide.st;
                tls_tmp = &a.wide.tmp;
                tls_arena = &ix->arena[tls_lane];
                wide_run(items, level, a.wide);
            } catch (...) {
                wide_err = std::current_exception();
            }
            tls_stream = nullptr;
            tls_tmp = nullptr;
        });
    };
    // the classes side by side, the long ones first (their wavefronts should all be resident from the start); the caller's
    // thread takes the shortest class on its own stream
    // every stream exists before the first thread starts: a failing hipStreamCreate must not unwind past joinable threads
    for (int c = NCH - 1; c >= 1; c--)
        if (!cls[c].empty() && !a.lean[c].st) HIPCHK(hipStreamCreate(&a.lean[c].st));
    if (!a.wide.st) HIPCHK(hipStreamCreate(&a.wide.st));
    std::thread cth[NCH];
    std::exception_ptr cerr[NCH];
    const bool serial = ix->tune.wfa_serial; // exclusive kernel timings: one class after the other
    // start order: the long classes first (latency-bound: resident from the start)
    int start_order[NCH];
    for (int c = 0; c < NCLS; c++) start_order[c] = NCLS - 1 - c;
    for (int oi = 0; oi < NCH; oi++) {
        const int c = start_order[oi];
        if (c == 0 || cls[c].empty()) continue;
        cth[c] = std::thread([&, c]() {
            try {
                HIPCHK(hipSetDevice(ix->device));
                tls_lane = lane;
                tls_stream = a.lean[c].st;
                tls_tmp = &a.lean[c].tmp;
                tls_arena = &ix->arena[tls_lane];
                class_chain(c);
            } catch (...) {
                cerr[c] = std::current_exception();
            }
            tls_stream = nullptr;
            tls_tmp = nullptr;
        });
        if (serial) cth[c].join();
    }
    std::exception_ptr err0;
    try {
        class_chain(0);
        for (int c = NCH - 1; c >= 2; c--)
            if (c != 1 && c != NCLS && c != NCLS + 1 && cth[c].joinable()) cth[c].join();
        start_wide(); // the long classes are done: their leftovers run beside what is left of the short classes
    } catch (...) {
        err0 = std::current_exception();
    }
    for (int c = NCH - 1; c >= 1; c--)
        if (cth[c].joinable()) cth[c].join();
    if (wide_thread.joinable()) wide_thread.join();
    if (err0) std::rethrow_exception(err0);
    for (int c = 1; c < NCH; c++)
        if (cerr[c]) std::rethrow_exception(cerr[c]);
    if (wide_err) std::rethrow_exception(wide_err);
    if (!fb_items.empty()) { // leftovers of the short classes (rare): same fallback, on this thread's stream
        wide_run(fb_items, fb_level, a.wide);
    }
    a.stats->wfa_retries += retries.load();
    if (want_ops) {
        for (int64_t i = 0; i < n; i++) {
            ops_off_h[i] = (int64_t)ops_h.size();
            ops_h.insert(ops_h.end(), ops_keep[i].begin(), ops_keep[i].end());
        }
        ops_off_h[n] = (int64_t)ops_h.size();
    }
}

// scoreAndEvalue (lib-index-search-util.go:260-304) from the kernel's integer score
static void score_evalue(int score, int qlen, int64_t total_bases, int *bitscore, double *evalue) {
    int _score = score;
    if ((_score & 1) == 1) _score--;
    const double lnK = std::log(0.41);
    double bit = (0.625 * (double)_score - lnK) / 0.693147180559945309417232121458176568;
    *evalue = (double)total_bases * std::pow(2, -bit) * (double)qlen;
    *bitscore = (int)bit;
}

static std::string *fmt_cigar(const std::vector<uint64_t> &ops) { // :2327-2340
    int start = -1, end = -1;
    for (size_t i = 0; i < ops.size(); i++)
        if ((ops[i] >> 32) == 'M') {
            if (start < 0) start = (int)i;
            end = (int)i;
        }
    std::string *s = new std::string();
    for (int i = start; i >= 0 && i <= end; i++) {
        char op = (char)(ops[i] >> 32);
        if (op == 'D')
            op = 'I';
        else if (op == 'I')
            op = 'D';
        *s += std::to_string((unsigned)(ops[i] & 0xffffffffu));
        *s += op;
    }
    return s;
}

static void fmt_alignment(const std::vector<uint64_t> &ops, const uint8_t *q, const uint8_t *t, std::string *Q,
                          std::string *A, std::string *T) {
    int start = -1, end = -1;
    for (size_t i = 0; i < ops.size(); i++)
        if ((ops[i] >> 32) == 'M') {
            if (start < 0) start = (int)i;
            end = (int)i;
        }
    int qp = 0, tp = 0;
    for (int i = 0; i < (int)ops.size(); i++) {
        char op = (char)(ops[i] >> 32);
        int cnt = (int)(ops[i] & 0xffffffffu);
        bool in = start >= 0 && i >= start && i <= end;
        for (int j = 0; j < cnt; j++) {
            if (op == 'M' || op == 'X') {
                if (in) {
                    Q->push_back((char)q[qp]);
                    T->push_back((char)t[tp]);
                    A->push_back(op == 'M' ? '|' : ' ');
                }
                qp++;
                tp++;
            } else if (op == 'I') {
                if (in) {
                    Q->push_back('-');
                    T->push_back((char)t[tp]);
                    A->push_back(' ');
                }
                tp++;
            } else {
                if (in) {
                    Q->push_back((char)q[qp]);
                    T->push_back('-');
                    A->push_back(' ');
                }
                qp++;
            }
        }
    }
}

static Work &get_work(lm_index *ix, lm_qbatch *qb) {
    Work *&wk = lane_work(ix);
    if (!wk) wk = new Work(ix, qb);
    wk->rebind(qb);
    return *wk;
}
static AlignCtx &get_actx(lm_index *ix, lm_qbatch *qb, Work *w, lm_stage_stats *st, int slot = 0) {
    AlignCtx **ac = lane_actx(ix);
    if (!ac[slot]) ac[slot] = new AlignCtx();
    AlignCtx &a = *ac[slot];
    a.ix = ix;
    a.qb = qb;
    a.w = w;
    a.stats = st;
    return a;
}

// The alignment half for tasks [r0, r1) of the batch (whole (query, genome) segments): pseudo-alignment -> glue ->
// extendMatch -> WFA -> per-genome finalisation, in chunks bounded by the window budget. Appends to `genomes` in task
// order. Runs on the calling thread's stream (tls_stream) with the private scratch of `a`.
static void align_range(lm_index *ix, lm_qbatch *qb, Work &w, AlignCtx &a, TaskSpan tasks_h, int64_t r0, int64_t r1,
                        lm_stage_stats &st, std::vector<HGenome> &genomes, lm_result *res, std::mutex &strings_mu) {
    // windows of one alignment chunk: the chunk's pseudo-alignment anchors (~0.04 per window base, ~90 B of scratch each)
    // and its WFA launches scale with it; long alignments are latency-bound per problem, so few large chunks beat many
    // small ones (each chunk ends with a tail of a few 50-kb alignments running alone)
    int64_t max_window_bytes = BUDGET(ix) > 0
                                   ? std::min<int64_t>((int64_t)16 << 30, std::max<int64_t>((int64_t)1 << 30, BUDGET(ix) * 5 / 100))
                                   : (int64_t)2 << 30;
    if (const char *e = getenv("LM_DEBUG_MAX_WINDOW_BYTES")) max_window_bytes = std::max<int64_t>(1, atoll(e)); // test hook
    const bool want_seq = ix->opt.output_seq != 0;
    // ---- two-stage pipeline over the chunks: a producer thread runs the pseudo-alignment of chunk c+1 on its own stream and
    // context while this thread takes chunk c through glue -> extendMatch -> WFA -> finalisation.  The WFA passes end in
    // tails of a few long alignments that leave most CUs idle; the anchor kernel of the next chunk fills them.
    struct PaChunk {
        int64_t tpos = 0, tend = 0, base = 0, off = 0;
        std::vector<int64_t> res_off;
        std::vector<LmChain2> resv;
        int slot = 0;
    };
    AlignCtx *ctxs[2] = {&a, &get_actx(ix, qb, &w, &st, 1)};
    ctxs[1]->pa_ratio = ctxs[0]->pa_ratio;
    std::mutex pm;
    std::condition_variable pcv;
    std::deque<PaChunk> ready;
    bool slot_free[2] = {true, true}, prod_done = false, cons_abort = false;
    std::exception_ptr prod_err;
    double ms_pseudo = 0;
    const int64_t total_window = r1 > r0 ? tasks_h[r1 - 1].woff + tasks_h[r1 - 1].wlen - tasks_h[r0].woff : 0;
    const bool pipelined = total_window > max_window_bytes && !ix->tune.no_pipeline;
    const int lane = tls_lane;
    auto producer = [&]() {
        try {
            if (pipelined) {
                HIPCHK(hipSetDevice(ix->device));
                tls_lane = lane;
                if (!lane_st2(ix)) HIPCHK(hipStreamCreate(&lane_st2(ix)));
                tls_stream = lane_st2(ix);
                tls_tmp = &lane_tmp2(ix);
                tls_arena = &ix->arena[tls_lane];
            }
            int64_t tpos = r0;
            int slot = 0;
            while (tpos < r1) {
                if (BUDGET(ix) > 0 && *ctxs[0]->pa_ratio > 0) { // expected anchors within 80 % of the chunk's share
                    const int64_t lim = (int64_t)((double)(BUDGET(ix) * 13 / 100) / 90.0 * 0.8 / *ctxs[0]->pa_ratio);
                    if (!getenv("LM_DEBUG_MAX_WINDOW_BYTES")) max_window_bytes = std::min(max_window_bytes, std::max<int64_t>(lim, 1 << 20));
                }
                // chunk [tpos, tend): whole segments, bounded window bytes
                int64_t tend = tpos, wb = 0;
                while (tend < r1) {
                    int64_t e = tend;
                    uint32_t seg = tasks_h[tend].seg;
                    int64_t segw = 0;
                    while (e < r1 && tasks_h[e].seg == seg) segw += tasks_h[e++].wlen;
                    if (tend > tpos && wb + segw > max_window_bytes) break;
                    wb += segw;
                    tend = e;
                }
                if (getenv("LM_DEBUG"))
                    fprintf(stderr, "[lm] alignment chunk: tasks [%lld, %lld) of [%lld, %lld), %.2f GB of windows (limit %.2f GB, %.4f anchors per window byte so far)\n",
                            (long long)tpos, (long long)tend, (long long)r0, (long long)r1, (double)wb / 1e9,
                            (double)max_window_bytes / 1e9, *ctxs[0]->pa_ratio);
                {
                    std::unique_lock<std::mutex> l(pm);
                    pcv.wait(l, [&] { return slot_free[slot] || cons_abort; });
                    if (cons_abort) break;
                }
                // host and device task lists are used in place: window offsets are global, this chunk's slice starts at `base`
                PaChunk pc;
                pc.tpos = tpos;
                pc.tend = tend;
                pc.slot = slot;
                TaskSpan ht;
                ht.p = tasks_h.p + tpos;
                ht.n = (size_t)(tend - tpos);
                pc.base = ht[0].woff;
                pc.off = ht.back().woff + ht.back().wlen - pc.base; // window bytes of the chunk
                const double ta = now_ms();
                dbg_stamp("pseudo-alignment of a chunk starts");
                try {
                    run_pseudo(*ctxs[slot], ht, pc.res_off, pc.resv, w.tasks.p + tpos, pc.base, false);
                } catch (const ChunkTooLarge &) { // same tasks again in smaller chunks
                    if (ht[0].seg == ht.back().seg) throw HipError("too many pseudo-alignment anchors for one (query, genome) pair");
                    max_window_bytes = std::max<int64_t>(pc.off / 2, 1);
                    if (getenv("LM_DEBUG"))
                        fprintf(stderr, "[lm] alignment chunk halved to %lld window bytes\n", (long long)max_window_bytes);
                    continue;
                }
                ms_pseudo += now_ms() - ta;
                dbg_stamp("pseudo-alignment of a chunk done (chains on the host)");
                {
                    std::lock_guard<std::mutex> l(pm);
                    slot_free[slot] = false;
                    ready.push_back(std::move(pc));
                }
                pcv.notify_all();
                slot ^= 1;
                tpos = tend;
                if (!pipelined) return; // single chunk at a time: the caller loops
            }
        } catch (...) {
            prod_err = std::current_exception();
        }
        if (pipelined) {
            tls_stream = nullptr;
            tls_tmp = nullptr;
        }
        {
            std::lock_guard<std::mutex> l(pm);
            prod_done = true;
        }
        pcv.notify_all();
    };
    std::thread prod_thread;
    if (pipelined) prod_thread = std::thread(producer);
    struct Joiner { // the producer never outlives this frame
        std::thread &t;
        std::mutex &m;
        std::condition_variable &cv;
        bool &abort;
        ~Joiner() {
            {
                std::lock_guard<std::mutex> l(m);
                abort = true;
            }
            cv.notify_all();
            if (t.joinable()) t.join();
        }
    } joiner{prod_thread, pm, pcv, cons_abort};
    // ---- rounds.  The HSPs gathered from one or more chunks go through extendMatch / WFA / finalisation together.  (Round 4
    // let a round's latency-bound alignments - the long classes and whatever outgrew its first pass's ring - finish on a third
    // context beside the NEXT round's first passes; the consumer then waited for the pseudo-alignment producer instead: C3 14.3
    // against 13.0 s per step.  Removed in round 6.)
    struct Round {
        std::vector<HspMeta> hsps;
        std::vector<HGenome> genomes; // of this round, in task order
        std::vector<WfaOut> wout;
        std::vector<uint64_t> ops_h;
        std::vector<int64_t> ops_off_h;
        std::vector<uint8_t> wbuf_h;
    };
    std::vector<std::unique_ptr<Round>> rounds_done; // in round order: their genomes are appended to `genomes` at the end
    std::unique_ptr<Round> cur(new Round());
    std::mutex st_mu; // the stage statistics (the finalisation is threaded)
    int round_no = 0;
    auto gwb = [&]() -> DBuf<uint8_t> & { return a.gwbuf; };
    int64_t gw_used = 0;        // bytes of the round's window buffer in use
    const int64_t gw_target = BUDGET(ix) > 0 ? std::min<int64_t>((int64_t)6 << 30, std::max<int64_t>((int64_t)64 << 20, BUDGET(ix) * 3 / 100)) : (int64_t)1 << 30;
    const int64_t round_hsps = getenv("LM_DEBUG_ROUND_HSPS") ? atoll(getenv("LM_DEBUG_ROUND_HSPS")) : 330000;
    // what an idle consumer aligns instead of waiting for the producer.  150 000 until the end of round 6: every round ends in the
    // latency-bound 512 / 1024-diagonal passes of a few long alignments (~130 ms per round at C3), and the idle rule made ~32
    // rounds per C3 step out of what ~20 full ones hold: 8.0 s per step against 7.54 with a full round only, 8.68 with 80 000
    // (profiles/r06_c3_ab_rounds.json, one index) - and a step whose number of rounds depended on thread timing.
    const int64_t min_round_hsps = getenv("LM_DEBUG_MIN_ROUND_HSPS") ? atoll(getenv("LM_DEBUG_MIN_ROUND_HSPS")) : round_hsps;
    // ---- finalisation of a round's genomes (:2266-2357 / :2533-2626, then :2684-2749); on the consumer's thread
    auto finalize_round = [&](Round &R) {
        const double td = now_ms();
        parallel_for((int64_t)R.genomes.size(), 128, [&](int64_t gb0, int64_t gb1) {
        for (size_t gi = (size_t)gb0; gi < (size_t)gb1; gi++) {
            HGenome &gen = R.genomes[gi];
            int qlen = (int)(qb->h_qoff[gen.q + 1] - qb->h_qoff[gen.q]);
            for (auto &cl : gen.sds) {
                double max_sim = 0;
                bool has = false;
                for (auto &c : cl.chains) {
                    if (!c.alive || c.hsp < 0) {
                        c.alive = false;
                        continue;
                    }
                    const HspMeta &h = R.hsps[c.hsp];
                    const WfaOut &wo = R.wout[c.hsp];
                    const LmWfaOut &cg = wo.r;
                    int lq = h.ext.qe - h.ext.qs, lt = h.ext.te - h.ext.ts;
                    c.score = wo.blast_score;
                    if (cg.status != 0) { // no 'M' op: scoreAndEvalue returns MaxFloat64 -> filtered by evalue
                        c.alive = false;
                        continue;
                    }
                    score_evalue(c.score, lq, ix->host.total_bases, &c.bitscore, &c.evalue);
                    if (c.evalue > ix->opt.max_evalue) {
                        c.alive = false;
                        continue;
                    }
                    c.qbegin -= h.ext.s1;
                    c.qend += h.ext.e1;
                    c.qbegin = c.qbegin + cg.qbegin - 1;
                    c.qend = c.qend - (lq - cg.qend);
                    if (cl.rc) {
                        c.tbegin -= h.ext.e2;
                        c.tend += h.ext.s2;
                        c.tbegin = c.tbegin + (lt - cg.tend);
                        if (cl.variant_a)
                            c.tend = c.tend - cg.tbegin - 1; // :2285 as written in the contig-switch copy
                        else
                            c.tend = c.tend - (cg.tbegin - 1); // :2552
                    } else {
                        c.tbegin -= h.ext.s2;
                        c.tend += h.ext.e2;
                        c.tbegin = c.tbegin + cg.tbegin - 1;
                        c.tend = c.tend - (lt - cg.tend);
                    }
                    c.aligned_bases_q = c.qend - c.qbegin + 1;
                    c.aligned_length = (int)cg.align_len;
                    c.matched_bases = (int)cg.matches;
                    c.gaps = (int)cg.gaps;
                    c.aligned_fraction = (double)c.aligned_bases_q / (double)qlen * 100;
                    if (c.aligned_fraction > 100) c.aligned_fraction = 100;
                    c.pident = (double)c.matched_bases / (double)cg.align_len * 100;
                    if (c.aligned_fraction < ix->opt.min_qcov_per_hsp || c.pident < ix->opt.align_min_pident) {
                        c.alive = false;
                        continue;
                    }
                    if (want_seq) {
                        std::vector<uint64_t> ops(R.ops_h.begin() + R.ops_off_h[c.hsp], R.ops_h.begin() + R.ops_off_h[c.hsp + 1]);
                        c.cigar = fmt_cigar(ops);
                        c.qseq = new std::string();
                        c.tseq = new std::string();
                        c.align = new std::string();
                        fmt_alignment(ops, qb->h_seq.data() + qb->h_qoff[h.q] + h.ext.qs,
                                      R.wbuf_h.data() + h.in.woff + h.ext.ts, c.qseq, c.align, c.tseq);
                        std::lock_guard<std::mutex> sl(strings_mu);
                        res->strings.push_back(c.cigar);
                        res->strings.push_back(c.qseq);
                        res->strings.push_back(c.tseq);
                        res->strings.push_back(c.align);
                    }
                    double sim = (double)c.bitscore * c.pident;
                    if (sim > max_sim) max_sim = sim;
                    has = true;
                }
                cl.has_result = has;
                cl.sim = max_sim;
            }
            // drop clusters without results (:2359-2391, :2628-2659)
            std::vector<HCluster> kept;
            for (auto &cl : gen.sds)
                if (cl.has_result) kept.push_back(std::move(cl));
            gen.sds.swap(kept);
            if (gen.sds.empty()) {
                gen.alive = false;
                continue;
            }
            std::vector<std::pair<int, int>> regions;
            for (auto &cl : gen.sds)
                for (auto &c : cl.chains)
                    if (c.alive) regions.push_back({c.qbegin, c.qend});
            int ab = coverage_len(regions);
            gen.aligned_fraction = (double)ab / (double)qlen * 100;
            if (gen.aligned_fraction > 100) gen.aligned_fraction = 100;
            // with split genomes in the index the filter waits for the chunk merge (:2701 "do not filter results now")
            if (!ix->host.has_chunks && gen.aligned_fraction < ix->opt.min_qcov_per_genome) {
                gen.alive = false;
                continue;
            }
            std::stable_sort(gen.sds.begin(), gen.sds.end(), [](const HCluster &x, const HCluster &y) { return x.sim > y.sim; });
        }
        });
        if (getenv("LM_DEBUG")) fprintf(stderr, "[lm] finalize: parallel part %.2f ms\n", now_ms() - td);
        {
            std::lock_guard<std::mutex> l(st_mu);
            st.ms_finalize += now_ms() - td;
        }
        janitor().dispose(std::move(R.hsps));
        janitor().dispose(std::move(R.wout));
        R.hsps = std::vector<HspMeta>();
        R.ops_h = std::vector<uint64_t>();
        R.wbuf_h = std::vector<uint8_t>();
    };
    auto flush_round = [&]() {
        if (cur->hsps.empty() && cur->genomes.empty()) {
            gw_used = 0;
            return;
        }
        Round &R = *cur;
        std::vector<HspMeta> &hsps = R.hsps;
        DBuf<uint8_t> &gwbuf = gwb();
        double tc = now_ms();
        int64_t NH = (int64_t)hsps.size();
        st.hsps_aligned += NH;
        dbg_stamp("extendMatch / WFA round starts");
        if (getenv("LM_DEBUG"))
            fprintf(stderr, "[lm] mem: round of %lld HSPs starts with %.2f GB of scratch held (budget %.2f)\n", (long long)NH,
                    (double)(g_dbuf_bytes.load() - ix->hbm_bytes) / 1e9, (double)BUDGET(ix) / 1e9);
        if (NH > 0) {
            std::vector<HspIn> hin(NH);
            for (int64_t i = 0; i < NH; i++) hin[i] = hsps[i].in;
            a.hsp_in.ensure((size_t)NH);
            a.hsp_ext.ensure((size_t)NH);
            a.ext_cap.ensure((size_t)NH + 1);
            a.ext_off.ensure((size_t)NH + 2);
            HIPCHK(hipMemcpyAsync(a.hsp_in.p, hin.data(), sizeof(HspIn) * NH, hipMemcpyHostToDevice, S(ix)));
            HIPCHK(hipMemsetAsync(a.ext_cap.p + NH, 0, sizeof(int32_t), S(ix)));
            launch_extend_count(S(ix), a.hsp_in.p, NH, qb->d_seq.p, qb->d_qoff.p, gwbuf.p, a.ext_cap.p);
            // scratch rows per wavefront of k_extend (32 HSPs = 64 flanks each), transposed layout
            const int64_t NW = (2 * NH + 63) / 64;
            a.ext_wcap.ensure((size_t)NW + 1);
            a.ext_off.ensure((size_t)NW + 2);
            HIPCHK(hipMemsetAsync(a.ext_wcap.p + NW, 0, sizeof(int32_t), S(ix)));
            launch_extend_wave_cap(S(ix), a.ext_cap.p, NH, a.ext_wcap.p, NW);
            int64_t ER = scan_to_i64<int32_t, CastI32>(ix, a.ext_wcap.p, NW, a.ext_off.p);
            a.ext_rows.ensure((size_t)extend_grid_blocks(NH) * LM_EXT_ROWS * 64 * 2);
            a.ext_rstart.ensure((size_t)extend_grid_blocks(NH) * LM_EXT_ROWS * 64);
            a.ext_subs.ensure(64 * (size_t)ER + 64);
            a.ext_msi.ensure(64 * (size_t)ER + 64);
            {
                // algorithmic bytes: per HSP the two flank pairs extendMatch reads (<= ext_len + 2 bases of query and of window on
                // either side, lib-index-search-util.go:34-201), its descriptor and its result
                Prof p(ix, "k_extend", NH * (int64_t)(4 * (ix->opt.ext_len2 + 2) + sizeof(HspIn) + sizeof(HspExt)));
                launch_extend(S(ix), a.hsp_in.p, NH, qb->d_seq.p, qb->d_qoff.p, gwbuf.p, a.ext_cap.p, a.ext_off.p,
                              a.ext_subs.p, a.ext_msi.p, a.ext_rows.p, a.ext_rstart.p, a.hsp_ext.p);
            }
            std::vector<HspExt> hext;
            d2h(ix, hext, a.hsp_ext.p, (size_t)NH);
            sync(ix);
            std::vector<WfaIn> win(NH);
            for (int64_t i = 0; i < NH; i++) {
                hsps[i].ext = hext[i];
                win[i].q = qb->d_seq.p + qb->h_qoff[hsps[i].q] + hext[i].qs;
                win[i].t = gwbuf.p + hsps[i].in.woff + hext[i].ts;
                win[i].qlen = hext[i].qe - hext[i].qs;
                win[i].tlen = hext[i].te - hext[i].ts;
            }
            std::vector<float> est(NH);
            for (int64_t i = 0; i < NH; i++) est[i] = hsps[i].est_div;
            run_wfa(a, win, R.wout, R.ops_h, R.ops_off_h, want_seq, &est);
            if (want_seq) {
                d2h(ix, R.wbuf_h, gwbuf.p, (size_t)gw_used);
                sync(ix);
            }
        }
        {
            std::lock_guard<std::mutex> l(st_mu);
            st.ms_extend_wfa += now_ms() - tc;
        }
        dbg_stamp("WFA passes of the round done");
        std::unique_ptr<Round> done = std::move(cur);
        cur.reset(new Round());
        round_no++;
        gw_used = 0;
        Round *Rp = done.get();
        rounds_done.push_back(std::move(done));
        finalize_round(*Rp);
    };
    int64_t np_tpos = r0; // unpipelined: next chunk start
    while (true) {
        PaChunk pc;
        if (pipelined) {
            std::unique_lock<std::mutex> l(pm);
            pcv.wait(l, [&] { return !ready.empty() || prod_done; });
            if (ready.empty()) break;
            pc = std::move(ready.front());
            ready.pop_front();
        } else {
            if (np_tpos >= r1) break;
            // one chunk, inline: same code path, same thread and stream
            const int64_t save_r0 = r0;
            r0 = np_tpos;
            prod_done = false;
            producer();
            r0 = save_r0;
            if (prod_err) std::rethrow_exception(prod_err);
            if (ready.empty()) break;
            pc = std::move(ready.front());
            ready.pop_front();
            np_tpos = pc.tend;
        }
        const int64_t tpos = pc.tpos;
        TaskSpan ht;
        ht.p = tasks_h.p + tpos;
        ht.n = (size_t)(pc.tend - tpos);
        std::vector<int64_t> &res_off = pc.res_off;
        std::vector<LmChain2> &resv = pc.resv;
        // this chunk's windows of tasks with chains must fit the round's buffer: close the round first if they might not
        {
            int64_t need = 0;
            for (size_t t = 0; t < ht.size(); t++)
                if (res_off[t + 1] > res_off[t]) need += ((int64_t)ht[t].wlen + 15) & ~(int64_t)15;
            if (gw_used > 0 && (gw_used + need > (int64_t)gwb().cap - 64 || (int64_t)cur->hsps.size() >= round_hsps)) flush_round();
            if (gw_used == 0) gwb().ensure((size_t)std::max<int64_t>(need, gw_target) + 64);
        }
        double tb = now_ms();
        dbg_stamp("glue of a chunk starts");
        // glue per segment with results (parallel; the order of `genomes` stays the segment order)
        std::vector<HspMeta> &hsps = cur->hsps;
        const size_t hs0 = hsps.size(); // this chunk's HSPs go behind the round's
        {
            std::vector<std::pair<size_t, size_t>> active; // task ranges of the segments that have Chain2 results
            for (size_t i = 0; i < ht.size();) {
                size_t e = i;
                while (e < ht.size() && ht[e].seg == ht[i].seg) e++;
                if (res_off[e] > res_off[i] && ht[i].g >= 0) active.push_back({i, e});
                i = e;
            }
            const int64_t ns = (int64_t)active.size();
            const double tg0 = now_ms();
            std::vector<HGenome> gens((size_t)ns);
            std::vector<int32_t> nh((size_t)ns + 1, 0); // HSPs per segment
            (void)div_from_pseudo_pident(0);            // builds its table before the threads use it
            parallel_for(ns, 64, [&](int64_t s0, int64_t s1) {
                for (int64_t si = s0; si < s1; si++) {
                    size_t i = active[si].first, e = active[si].second;
                    HGenome &gen = gens[si];
                    gen.q = ht[i].q;
                    gen.bg = ht[i].bg;
                    gen.g = ht[i].g;
                    std::map<AKey, bool> keys;
                    for (size_t t = i; t < e; t++)
                        glue_task(ix, gen, keys, ht[t], (int64_t)t, resv.data() + res_off[t],
                                  (int)(res_off[t + 1] - res_off[t]));
                    // HSP list (Update2 + the start of the finalisation loops :2223-2255 / :2490-2522): which chains
                    // go to extendMatch/WFA; c.hsp = index within the genome for now
                    int qlen = (int)(qb->h_qoff[gen.q + 1] - qb->h_qoff[gen.q]);
                    int cnt = 0;
                    for (auto &cl : gen.sds)
                        for (auto &c : cl.chains) {
                            c.aligned_fraction = (double)c.aligned_bases_q / (double)qlen * 100;
                            if (c.qbegin >= c.qend + 1) {
                                c.alive = false;
                                continue;
                            }
                            int start, end;
                            if (cl.rc) {
                                start = cl.tEnd - c.tend - c.tpos_offset_begin;
                                end = cl.tEnd - c.tbegin - c.tpos_offset_begin + 1;
                            } else {
                                start = c.tpos_offset_begin + c.tbegin - cl.tBegin;
                                end = c.tpos_offset_begin + c.tend - cl.tBegin + 1;
                            }
                            if (start >= end) {
                                c.alive = false;
                                continue;
                            }
                            c.hsp = cnt++;
                        }
                    nh[si] = cnt;
                }
            });
            const double tg1 = now_ms();
            std::vector<int64_t> hbase((size_t)ns + 1, 0);
            for (int64_t si = 0; si < ns; si++) hbase[si + 1] = hbase[si] + nh[si];
            hsps.resize(hs0 + (size_t)hbase[ns]);
            // compact window offsets (in the round's window buffer) of the tasks with chains
            std::vector<int64_t> tdest(ht.size(), -1);
            std::vector<int32_t> widx;
            std::vector<int64_t> wdest;
            for (int64_t si = 0; si < ns; si++)
                for (size_t t = active[si].first; t < active[si].second; t++)
                    if (res_off[t + 1] > res_off[t] && ht[t].wlen > 0) {
                        tdest[t] = gw_used;
                        widx.push_back((int32_t)t);
                        wdest.push_back(gw_used);
                        gw_used += ((int64_t)ht[t].wlen + 15) & ~(int64_t)15;
                    }
            parallel_for(ns, 64, [&](int64_t s0, int64_t s1) {
                for (int64_t si = s0; si < s1; si++) {
                    HGenome &gen = gens[si];
                    int qlen = (int)(qb->h_qoff[gen.q + 1] - qb->h_qoff[gen.q]);
                    for (auto &cl : gen.sds) {
                        const Task &t = ht[cl.task];
                        for (auto &c : cl.chains) {
                            if (!c.alive || c.hsp < 0) continue;
                            int start, end;
                            if (cl.rc) {
                                start = cl.tEnd - c.tend - c.tpos_offset_begin;
                                end = cl.tEnd - c.tbegin - c.tpos_offset_begin + 1;
                            } else {
                                start = c.tpos_offset_begin + c.tbegin - cl.tBegin;
                                end = c.tpos_offset_begin + c.tend - cl.tBegin + 1;
                            }
                            int ext2 = ix->opt.ext_len2;
                            if (c.aligned_bases_q > 1000000)
                                ext2 += 80;
                            else if (c.aligned_bases_q > 250000)
                                ext2 += 40;
                            else if (c.aligned_bases_q > 50000)
                                ext2 += 20;
                            else if (c.aligned_bases_q > 10000)
                                ext2 += 10;
                            c.hsp += (int64_t)hs0 + hbase[si];
                            HspMeta &h = hsps[(size_t)c.hsp];
                            h.task = cl.task;
                            h.est_div = (float)div_from_pseudo_pident(c.pident);
                            h.q = gen.q;
                            h.in.q = gen.q;
                            h.in.rc = cl.rc ? 1 : 0;
                            h.in.woff = tdest[cl.task];
                            h.in.len1 = qlen;
                            h.in.len2 = t.wlen;
                            h.in.start1 = c.qbegin;
                            h.in.end1 = c.qend + 1;
                            h.in.start2 = start;
                            h.in.end2 = end;
                            h.in.ext_len = ext2;
                            h.in.tbegin = c.tbegin;
                            h.in.max_ext_len = c.max_ext_len;
                            h.in.pad = 0;
                        }
                    }
                }
            });
            const double tg2 = now_ms();
            // the windows of those tasks, from the 2-bit genomes into the round's buffer (it was sized before this chunk)
            if (!widx.empty()) {
                a.gw_idx.ensure(widx.size());
                a.gw_dest.ensure(wdest.size());
                HIPCHK(hipMemcpyAsync(a.gw_idx.p, widx.data(), widx.size() * sizeof(int32_t), hipMemcpyHostToDevice, S(ix)));
                HIPCHK(hipMemcpyAsync(a.gw_dest.p, wdest.data(), wdest.size() * sizeof(int64_t), hipMemcpyHostToDevice, S(ix)));
                Prof p(ix, "k_extract_windows", (gw_used - wdest[0]) * 5 / 4);
                launch_extract_windows_at(S(ix), ix->view, w.tasks.p + tpos, a.gw_idx.p, a.gw_dest.p, (int64_t)widx.size(), gwb().p);
                sync(ix); // the host lists go out of scope
            }
            cur->genomes.reserve(cur->genomes.size() + (size_t)ns);
            for (int64_t si = 0; si < ns; si++)
                if (!gens[si].sds.empty()) cur->genomes.push_back(std::move(gens[si]));
            if (getenv("LM_DEBUG"))
                fprintf(stderr, "[lm] glue: active scan %.2f, glue_task pass %.2f, hsp fill %.2f, move %.2f ms (%lld genomes)\n",
                        tg0 - tb, tg1 - tg0, tg2 - tg1, now_ms() - tg2, (long long)ns);
        }
        {
            std::lock_guard<std::mutex> l(st_mu);
            st.ms_glue += now_ms() - tb;
        }
        dbg_stamp("glue of a chunk done");
        janitor().dispose(std::move(resv));
        bool idle = false;
        {   // the chunk's pseudo-alignment results are consumed and its windows copied: its context may take the next chunk
            std::lock_guard<std::mutex> l(pm);
            slot_free[pc.slot] = true;
            idle = pipelined && ready.empty() && !prod_done;
        }
        pcv.notify_all();
        // nothing to consume yet: align what has been gathered instead of waiting for the producer (the anchor kernels of the
        // next chunks then run beside the WFA launches, whose scalar-unit-bound wavefronts leave the vector ALUs and LDS idle)
        if (idle && (int64_t)cur->hsps.size() >= min_round_hsps) flush_round();
    }
    flush_round();
    {
        size_t total = genomes.size();
        for (auto &r : rounds_done) total += r->genomes.size();
        genomes.reserve(total);
        for (auto &r : rounds_done)
            for (auto &g : r->genomes) genomes.push_back(std::move(g));
        rounds_done.clear();
    }
    if (pipelined) {
        if (prod_thread.joinable()) prod_thread.join();
        if (prod_err) std::rethrow_exception(prod_err);
    }
    st.ms_pseudo += ms_pseudo;
}

// what a search pass is asked to do besides the plain search: report the per-(query, genome) chaining scores and stop
// (lm_search_scores), or replace the local -n cut by a list of (query, genome) pairs to keep (lm_search_resident_keep)
struct SearchCtl {
    std::vector<uint32_t> *sc_query = nullptr;
    std::vector<uint64_t> *sc_bg = nullptr;
    std::vector<float> *sc_score = nullptr;
    const std::unordered_map<uint64_t, std::vector<uint32_t>> *keep = nullptr; // genome key -> sorted batch query numbers
};

static void search_impl(lm_index *ix, lm_qbatch *qb, lm_result *res, const SearchCtl *ctl = nullptr) {
    // the caller (search_parts) holds the handle's mutex and has set this thread's lane
    tune_malloc_once();
    lm_stage_stats &st = res->stats;
    memset(&st, 0, sizeof st);
    HIPCHK(hipSetDevice(ix->device));
    double t0 = now_ms(), t1;
    st.query_bases = qb->total_len;
    st.query_kmers = 2 * qb->total_pos;
    struct TlsScope { // allocations and launches of this thread belong to this handle's arena and stream
        ScratchArena *pa = tls_arena;
        hipStream_t ps = tls_stream;
        explicit TlsScope(lm_index *ix) {
            tls_arena = &ix->arena[tls_lane];
            if (!tls_stream) tls_stream = lane_st(ix);
        }
        ~TlsScope() {
            tls_arena = pa;
            tls_stream = ps;
        }
    } tls_scope(ix);
    g_dbg_t0 = now_ms();
    dbg_stamp("search of a batch part starts");
    if (BUDGET(ix) > 0) { // scratch of the previous part's alignment half (the halves alternate)
        HIPCHK(hipStreamSynchronize(S(ix))); // (every worker thread of the previous part synchronised its stream and was joined)
        int64_t freed = 0;
        for (int j = 0; j < 3; j++)
            if (AlignCtx *c = lane_actx(ix)[j]) freed += c->release_big(BUDGET(ix) / 200);
        if (freed > 0 && getenv("LM_DEBUG"))
            fprintf(stderr, "[lm] alignment scratch of the previous part released: %.2f GB (arena: %.2f GB in slabs, %lld slab allocations so far)\n",
                    (double)freed / 1e9, (double)ix->arena[tls_lane].slab_bytes / 1e9, (long long)ix->arena[tls_lane].slab_allocs);
    }
    dbg_stamp("previous alignment scratch released");
    Work &w = get_work(ix, qb);
    double tm1 = now_ms();
    stage_kmers(w);
    double tm2 = now_ms();
    stage_mask(w);
    double tm3 = now_ms();
    sync(ix);
    t1 = now_ms();
    if (getenv("LM_DEBUG"))
        fprintf(stderr, "[lm] mask stage: get_work %.2f, kmers(launch) %.2f, mask(launch) %.2f, sync %.2f ms\n", tm1 - t0,
                tm2 - tm1, tm3 - tm2, t1 - tm3);
    st.ms_mask = t1 - t0;
    t0 = t1;
    stage_lookup(w, st);
    t1 = now_ms();
    st.ms_lookup = t1 - t0;
    t0 = t1;
    if (w.nseg == 0) {
        st.ms_total = st.ms_mask + st.ms_lookup;
        return;
    }
    stage_chain1(w);
    int nseg = w.nseg;
    // ---- top-N genomes per query (lib-index-search.go:1781-1805), host selection on (score desc, genome asc)
    std::vector<uint64_t> segA_h;
    std::vector<float> score_h;
    {   // anchors surviving ClearSubstrPairs, summed on the device (statistics only)
        unsigned long long hv = 0;
        HIPCHK(hipMemsetAsync(w.stat.p + 1, 0, sizeof(unsigned long long), S(ix)));
        launch_sum_i32(S(ix), w.seg_n.p, nseg, w.stat.p + 1);
        HIPCHK(hipMemcpyAsync(&hv, w.stat.p + 1, sizeof hv, hipMemcpyDeviceToHost, S(ix)));
        if (ix->opt.top_n_genomes > 0 || ctl) { // only the top-N selection needs the per-pair scores on the host
            d2h(ix, segA_h, w.segA.p, (size_t)nseg);
            d2h(ix, score_h, w.seg_score.p, (size_t)nseg);
        }
        sync(ix);
        st.anchors_cleared += (int64_t)hv;
    }
    const float min_score = chain_opt(ix).min_score;
    const uint8_t *keep_d = nullptr;
    if (ctl && ctl->keep) { // the caller's (query, genome) list replaces the local cut
        std::vector<uint8_t> keep(nseg, 0);
        for (int i = 0; i < nseg; i++) {
            auto it = ctl->keep->find(segA_h[i] & ((1ull << 34) - 1));
            if (it == ctl->keep->end()) continue;
            const uint32_t q = (uint32_t)(segA_h[i] >> 34) + qb->q0;
            keep[i] = std::binary_search(it->second.begin(), it->second.end(), q) ? 1 : 0;
        }
        w.keep.ensure(nseg);
        HIPCHK(hipMemcpyAsync(w.keep.p, keep.data(), nseg, hipMemcpyHostToDevice, S(ix)));
        keep_d = w.keep.p;
    } else if (ix->opt.top_n_genomes > 0 || (ctl && ctl->sc_query)) {
        std::vector<uint8_t> keep(nseg, 0);
        int s = 0;
        while (s < nseg) {
            int e = s;
            uint64_t q = segA_h[s] >> 34;
            while (e < nseg && (segA_h[e] >> 34) == q) e++;
            std::vector<int> cand;
            for (int i = s; i < e; i++)
                if (score_h[i] >= min_score) cand.push_back(i);
            std::stable_sort(cand.begin(), cand.end(), [&](int x, int y) { return score_h[x] > score_h[y]; });
            const size_t lim = ix->opt.top_n_genomes > 0 ? (size_t)ix->opt.top_n_genomes : cand.size();
            for (size_t i = 0; i < cand.size() && i < lim; i++) {
                keep[cand[i]] = 1;
                if (ctl && ctl->sc_query) { // report only
                    ctl->sc_query->push_back((uint32_t)(segA_h[cand[i]] >> 34) + qb->q0);
                    ctl->sc_bg->push_back(segA_h[cand[i]] & ((1ull << 34) - 1));
                    ctl->sc_score->push_back(score_h[cand[i]]);
                }
            }
            s = e;
        }
        if (ctl && ctl->sc_query) {
            st.ms_total = st.ms_mask + st.ms_lookup;
            return;
        }
        w.keep.ensure(nseg);
        HIPCHK(hipMemcpyAsync(w.keep.p, keep.data(), nseg, hipMemcpyHostToDevice, S(ix)));
        keep_d = w.keep.p;
    }
    w.ntask.ensure((size_t)nseg + 1);
    w.task_off.ensure((size_t)nseg + 2);
    HIPCHK(hipMemsetAsync(w.ntask.p + nseg, 0, sizeof(int32_t), S(ix)));
    launch_task_count(S(ix), w.seg_score.p, w.seg_nch.p, keep_d, nseg, min_score, w.ntask.p);
    int64_t NT = scan_to_i64<int32_t, CastI32>(ix, w.ntask.p, nseg, w.task_off.p);
    st.chains += NT;
    t1 = now_ms();
    st.ms_chain = t1 - t0;
    t0 = t1;
    if (NT == 0) {
        st.ms_total = st.ms_mask + st.ms_lookup + st.ms_chain;
        return;
    }
    w.tasks.ensure((size_t)NT);
    launch_make_tasks(S(ix), ix->view, w.segA.p, w.seg_off.p, nseg, w.subs.p, w.chain_off_pool.p, w.chain_idx_pool.p,
                      w.ntask.p, w.task_off.p, qb->d_qoff.p, ix->opt.ext_len, w.order_scratch.p, w.tasks.p);
    {   // window offsets of all tasks laid end to end (what a single alignment chunk uses as is)
        w.task_wlen.ensure((size_t)NT + 1);
        w.task_woff.ensure((size_t)NT + 2);
        HIPCHK(hipMemsetAsync(w.task_wlen.p + NT, 0, sizeof(int32_t), S(ix)));
        launch_task_wlen(S(ix), w.tasks.p, NT, w.task_wlen.p);
        (void)scan_to_i64<int32_t, CastI32>(ix, w.task_wlen.p, NT, w.task_woff.p);
        launch_task_set_woff(S(ix), w.tasks.p, NT, w.task_woff.p);
    }
    w.tasks_host.ensure((size_t)NT);
    Task *tasks_h = w.tasks_host.p; // pinned, reused across batches
    HIPCHK(hipMemcpyAsync(tasks_h, w.tasks.p, sizeof(Task) * (size_t)NT, hipMemcpyDeviceToHost, S(ix)));
    sync(ix);
    t1 = now_ms();
    st.ms_window = t1 - t0;
    t0 = t1;

    if (BUDGET(ix) > 0) {
        const int64_t before = g_dbuf_bytes.load();
        HIPCHK(hipStreamSynchronize(S(ix)));
        w.release_seeding(BUDGET(ix) / 200);
        if (getenv("LM_DEBUG"))
            fprintf(stderr, "[lm] mem: seeding half held %.2f GB of scratch (budget %.2f), %.2f GB kept for the alignment half\n",
                    (double)(before - ix->hbm_bytes) / 1e9, (double)BUDGET(ix) / 1e9,
                    (double)(g_dbuf_bytes.load() - ix->hbm_bytes) / 1e9);
    }
    // ---- alignment half, in chunks of whole (query, genome) segments (align_range). Splitting it over two host
    // threads / streams so that one half's host glue overlaps the other half's kernels was measured at C2 and gave
    // nothing (the halves run in lock-step, and the kernels only slow each other down), so it runs on one stream.
    dbg_stamp("seeding half done, tasks on the host");
    if (const char *e = getenv("LM_DEBUG_OOM_ABOVE_QUERIES")) // test hook: the out-of-memory answer of search_parts
        if ((int64_t)qb->nq > atoll(e)) throw DeviceOOM("test hook: allocation failure between the seeding and alignment halves");
    std::vector<HGenome> genomes; // in (query, genome) order
    {
        TaskSpan th;
        th.p = tasks_h;
        th.n = (size_t)NT;
        std::mutex strings_mu;
        AlignCtx &a = get_actx(ix, qb, &w, &st);
        align_range(ix, qb, w, a, th, 0, NT, st, genomes, res, strings_mu);
    }
    dbg_stamp("alignment half done");
    // ---- per query: sort genomes by best cluster (:2919-2921, ties by genome key), regroup by sseqid, emit rows ----
    double te0 = now_ms();
    {
        std::vector<size_t> qstart; // first genome of every query that has any
        for (size_t i = 0; i < genomes.size(); i++)
            if (i == 0 || genomes[i].q != genomes[i - 1].q) qstart.push_back(i);
        qstart.push_back(genomes.size());
        const int64_t nqg = (int64_t)qstart.size() - 1;
        std::vector<std::vector<lm_hsp>> qrows((size_t)nqg);
        std::vector<int64_t> qbases((size_t)nqg, 0);
        parallel_for(nqg, 4, [&](int64_t q0, int64_t q1) {
            for (int64_t qi = q0; qi < q1; qi++) {
                const size_t i = qstart[qi], e = qstart[qi + 1];
                std::vector<lm_hsp> &rows = qrows[qi];
                std::vector<HGenome *> gs;
                if (ix->host.has_chunks) {
                    // results of the chunks of one split genome are merged into the first of them (:2798-2851; the list is
                    // in genome order here), then EVERY result gets its query coverage, the -Q filter and the cluster
                    // order again (:2853-2897)
                    for (size_t j = i; j < e; j++) {
                        if (!genomes[j].alive) continue;
                        auto cj = ix->host.chunk_of.find(genomes[j].bg);
                        if (cj == ix->host.chunk_of.end()) continue;
                        for (size_t k2 = j + 1; k2 < e; k2++) {
                            if (!genomes[k2].alive) continue;
                            auto ck = ix->host.chunk_of.find(genomes[k2].bg);
                            if (ck == ix->host.chunk_of.end() || ck->second.list != cj->second.list) continue;
                            for (auto &cl : genomes[k2].sds) genomes[j].sds.push_back(std::move(cl));
                            genomes[k2].sds.clear();
                            genomes[k2].alive = false;
                        }
                    }
                    const int qlen = (int)(qb->h_qoff[genomes[i].q + 1] - qb->h_qoff[genomes[i].q]);
                    for (size_t j = i; j < e; j++) {
                        HGenome &gen = genomes[j];
                        if (!gen.alive) continue;
                        std::vector<std::pair<int, int>> regions;
                        for (auto &cl : gen.sds)
                            for (auto &c : cl.chains)
                                if (c.alive) regions.push_back({c.qbegin, c.qend});
                        gen.aligned_fraction = (double)coverage_len(regions) / (double)qlen * 100;
                        if (gen.aligned_fraction > 100) gen.aligned_fraction = 100;
                        if (gen.aligned_fraction < ix->opt.min_qcov_per_genome) {
                            gen.alive = false;
                            continue;
                        }
                        std::stable_sort(gen.sds.begin(), gen.sds.end(), [](const HCluster &x, const HCluster &y) { return x.sim > y.sim; });
                    }
                }
                for (size_t j = i; j < e; j++)
                    if (genomes[j].alive) gs.push_back(&genomes[j]);
                std::stable_sort(gs.begin(), gs.end(), [](const HGenome *x, const HGenome *y) {
                    if (x->sds[0].sim != y->sds[0].sim) return x->sds[0].sim > y->sds[0].sim;
                    return x->bg < y->bg;
                });
                std::vector<HCluster *> order;
                std::vector<char> used;
                for (HGenome *g : gs) {
                    const HostGenome &G0 = ix->host.genomes[g->g]; // the printed genome id: the (first) chunk's = the genome's
                    // SortBySeqID (:1042-1096): group clusters by sseqid keeping first-seen order
                    order.clear();
                    used.assign(g->sds.size(), 0);
                    for (size_t x = 0; x < g->sds.size(); x++) {
                        if (used[x]) continue;
                        for (size_t y = x; y < g->sds.size(); y++)
                            if (!used[y] && ((g->sds[y].g == g->sds[x].g && g->sds[y].seq_idx == g->sds[x].seq_idx) ||
                                             ix->host.genomes[g->sds[y].g].seq_ids[g->sds[y].seq_idx] ==
                                                 ix->host.genomes[g->sds[x].g].seq_ids[g->sds[x].seq_idx])) {
                                order.push_back(&g->sds[y]);
                                used[y] = 1;
                            }
                    }
                    int cls = 1, hspn = 1;
                    for (HCluster *cl : order) {
                        const HostGenome &G = ix->host.genomes[cl->g]; // contig table of the cluster's own chunk
                        for (auto &c : cl->chains) {
                            if (!c.alive) continue;
                            rows.emplace_back();
                            lm_hsp &r = rows.back();
                            memset(&r, 0, sizeof r);
                            r.query = g->q;
                            r.hits = (uint32_t)gs.size();
                            r.batch_genome = g->bg;
                            r.qcov_genome = g->aligned_fraction;
                            r.cls = cls;
                            r.hsp = hspn++;
                            r.seq_idx = cl->seq_idx;
                            r.nseqs = G.nseqs;
                            r.seq_len = G.seq_sizes[cl->seq_idx];
                            r.nchunks = cl->nchunks;
                            r.chunk_idx = cl->chunk_idx;
                            r.rc = cl->rc ? 1 : 0;
                            r.qcov_hsp = c.aligned_fraction;
                            r.aligned_length = c.aligned_length;
                            r.pident = c.pident;
                            r.gaps = c.gaps;
                            r.qbegin = c.qbegin;
                            r.qend = c.qend;
                            r.tbegin = c.tbegin;
                            r.tend = c.tend;
                            r.evalue = c.evalue;
                            r.bitscore = c.bitscore;
                            r.score = c.score;
                            r.matched_bases = c.matched_bases;
                            r.genome_id = G0.id.c_str();
                            r.seq_id = G.seq_ids[cl->seq_idx].c_str();
                            r.cigar = c.cigar ? c.cigar->c_str() : nullptr;
                            r.qseq = c.qseq ? c.qseq->c_str() : nullptr;
                            r.sseq = c.tseq ? c.tseq->c_str() : nullptr;
                            r.align = c.align ? c.align->c_str() : nullptr;
                            qbases[qi] += c.aligned_length;
                        }
                        cls++;
                    }
                }
            }
        });
        std::vector<size_t> roff((size_t)nqg + 1, 0);
        for (int64_t qi = 0; qi < nqg; qi++) {
            roff[qi + 1] = roff[qi] + qrows[qi].size();
            st.aligned_bases += qbases[qi];
        }
        res->rows.resize(roff[nqg]);
        st.rows += (int64_t)roff[nqg];
        parallel_for(nqg, 16, [&](int64_t q0, int64_t q1) {
            for (int64_t qi = q0; qi < q1; qi++)
                if (!qrows[qi].empty())
                    memcpy(res->rows.data() + roff[qi], qrows[qi].data(), qrows[qi].size() * sizeof(lm_hsp));
        });
    }
    if (getenv("LM_DEBUG")) fprintf(stderr, "[lm] row emission %.2f ms\n", now_ms() - te0);
    st.ms_finalize += now_ms() - te0;
    st.ms_total = st.ms_mask + st.ms_lookup + st.ms_chain + st.ms_window + st.ms_pseudo + st.ms_glue + st.ms_extend_wfa +
                  st.ms_finalize;
    janitor().dispose(std::move(genomes));
    dbg_stamp("rows emitted");
}

} // namespace lm

// two new batch parts from the host copy of the bases of `src` (a part, or the plain batch itself)
// A part thrown back because its seed anchors are `over` times what one pass may hold is cut into ceil(1.15 x over) pieces of
// about equal numbers of query BASES at once: halving it again and again searched the seeding stages of the same queries up to
// three times over in the first step of a fresh C3 batch (11.3 s where the settled step takes 9.7).
static std::vector<lm_qbatch *> split_qbatch(lm_index *ix, lm_qbatch *src, double over) {
    const size_t nq = (size_t)src->nq;
    if (nq < 2) throw HipError("one query yields more seed anchors than the device can hold");
    size_t k = (size_t)std::ceil(std::max(1.0, over) * 1.15);
    k = std::max<size_t>(2, std::min(k, nq));
    std::vector<lm_query> qs(nq);
    int64_t bases = 0;
    for (size_t i = 0; i < nq; i++) {
        qs[i].seq = src->h_seq.data() + src->h_qoff[i];
        qs[i].len = (uint32_t)(src->h_qoff[i + 1] - src->h_qoff[i]);
        bases += qs[i].len;
    }
    std::vector<lm_qbatch *> out;
    try {
        size_t b = 0;
        int64_t acc = 0; // bases of the queries before b
        for (size_t piece = 0; piece < k; piece++) {
            size_t e = b + 1; // every piece takes at least one query and leaves one for each piece behind it
            int64_t a2 = acc + qs[b].len;
            const int64_t want = bases * (int64_t)(piece + 1) / (int64_t)k;
            while (piece + 1 < k && e < nq - (k - 1 - piece) && a2 + qs[e].len / 2 <= want) a2 += qs[e++].len;
            if (piece + 1 == k) e = nq;
            out.push_back(upload_part(ix, qs.data() + b, e - b, src->q0 + (uint32_t)b));
            for (size_t i = b; i < e; i++) acc += qs[i].len;
            b = e;
        }
    } catch (...) {
        for (auto *p : out) delete p;
        throw;
    }
    if (getenv("LM_DEBUG")) fprintf(stderr, "[lm] batch part of %zu queries cut into %zu (seed anchors %.2f x the scratch budget's share)\n", nq, out.size(), over);
    return out;
}
static std::pair<lm_qbatch *, lm_qbatch *> halve_qbatch(lm_index *ix, lm_qbatch *src) {
    const size_t nq = (size_t)src->nq, h = nq / 2;
    if (nq < 2) throw HipError("one query yields more seed anchors than the device can hold");
    std::vector<lm_query> qs(nq);
    for (size_t i = 0; i < nq; i++) {
        qs[i].seq = src->h_seq.data() + src->h_qoff[i];
        qs[i].len = (uint32_t)(src->h_qoff[i + 1] - src->h_qoff[i]);
    }
    lm_qbatch *a = upload_part(ix, qs.data(), h, src->q0);
    lm_qbatch *b = nullptr;
    try {
        b = upload_part(ix, qs.data() + h, nq - h, src->q0 + (uint32_t)h);
    } catch (...) {
        delete a;
        throw;
    }
    if (getenv("LM_DEBUG")) fprintf(stderr, "[lm] batch part of %zu queries halved (seed anchors above the scratch budget)\n", nq);
    return {a, b};
}

extern "C" {

// the scratch of the lane this thread works for goes back to the device (after an allocation failure)
static void drop_scratch(lm_index *ix, const char *why) {
    (void)hipDeviceSynchronize();
    delete lane_work(ix);
    lane_work(ix) = nullptr;
    lm_free_align_ctx(ix, tls_lane);
    lane_tmp(ix).release();
    lane_tmp2(ix).release();
    ix->arena[tls_lane].trim(); // (its overflow slabs; the handle's lane slabs stay)
    if (getenv("LM_DEBUG")) fprintf(stderr, "[lm] device scratch of lane %d dropped after: %s\n", tls_lane, why);
}
// All parts of the caller's batch, results in order.  A part whose seed anchors outgrow the device (or whose scratch
// allocation fails) is halved in place; the split stays in the batch handle, so the next search of the same resident batch
// does not repeat it.  A batch of several parts is searched on two lanes (two host threads, each with its own scratch,
// streams and half of the scratch budget): the seeding and anchor kernels of one part (vector ALU, LDS, memory) run beside
// the WFA launches of another (bound by the scalar unit), and the host work of one beside the kernels of the other.
static void search_parts(lm_index *ix, lm_qbatch *qb, lm_result *res, const SearchCtl *ctl) {
    std::lock_guard<std::mutex> lock(ix->mu); // one in-flight call per handle
    HIPCHK(hipSetDevice(ix->device));
    tls_lane = 0;
    ix->active_lanes = 1;
    ix->budget_lanes = 1;
    lm_reserve_lane_slabs(ix); // once per handle (LaneSlabs, lm_internal.h); a production-size index did it when it was opened
    if (qb->parts.empty()) {
        ix->lane_slabs.assign(ix->arena[0], ix->arena[1], 1); // (false: a block is live - the assignment stays, overflow slabs serve)
        double over = 0;
        try {
            search_impl(ix, qb, res, ctl);
            return;
        } catch (const PartTooLarge &e) {
            over = e.over;
        } catch (const DeviceOOM &e) { // the shares of the scratch budget are estimates: retry on half the queries
            if (qb->nq < 2) throw;
            drop_scratch(ix, e.what());
        }
        // what the aborted attempt left in the caller's result (rows, CIGAR / alignment strings) must not outlive it
        for (auto *str : res->strings) delete str;
        res->strings.clear();
        res->rows.clear();
        res->stats = lm_stage_stats();
        if (over > 0) {
            qb->parts = split_qbatch(ix, qb, over);
        } else {
            auto ab = halve_qbatch(ix, qb);
            qb->parts = {ab.first, ab.second};
        }
        qb->d_seq.release(); // the plain batch's own device copy is no longer used
        qb->d_qoff.release();
        qb->d_posoff.release();
        qb->d_segoff.release();
    }
    struct PartState {
        lm_qbatch *part;
        int state; // 0 pending, 1 running, 2 done
        lm_result res;
    };
    std::list<PartState> todo;
    for (lm_qbatch *p : qb->parts) {
        todo.emplace_back();
        todo.back().part = p;
        todo.back().state = 0;
    }
    std::mutex lm_;
    std::exception_ptr err;
    // (round 3 measured +3..6 % at C3 and left it opt-in; with this round's WFA kernels the second lane is worth 10 %:
    // 12.97 -> 11.75 s per C3 step - the latency-bound passes of one part's rounds run beside the throughput-bound kernels of
    // the other part instead of beside their own round's; LM_TWO_LANES=0 turns it off)
    // (exclusive kernel timings: ONE lane works, but with the budget share, the slab and therefore the parts, chunks and
    // launches of the two-lane search it stands for)
    const int blanes = (ctl == nullptr && qb->parts.size() >= 2 && ix->tune.two_lanes) ? 2 : 1;
    const int lanes = (blanes == 2 && !ix->tune.wfa_serial) ? 2 : 1;
    ix->active_lanes = lanes;
    ix->budget_lanes = blanes;
    if (!ix->lane_slabs.assign(ix->arena[0], ix->arena[1], blanes)) { // a block outlived the previous search: give everything back first
        for (int l = 0; l < 2; l++) {
            tls_lane = l;
            drop_scratch(ix, "lane slabs change hands");
        }
        tls_lane = 0;
        ix->lane_slabs.assign(ix->arena[0], ix->arena[1], blanes);
    }
    auto lane_fn = [&](int lane) {
        tls_lane = lane;
        hipStream_t saved_stream = tls_stream;
        DBuf<uint8_t> *saved_tmp = tls_tmp;
        try {
            HIPCHK(hipSetDevice(ix->device));
            if (!lane_st(ix)) HIPCHK(hipStreamCreate(&lane_st(ix)));
            tls_stream = lane_st(ix);
            tls_tmp = &lane_tmp(ix);
            while (true) {
                std::list<PartState>::iterator it;
                {
                    std::lock_guard<std::mutex> l(lm_);
                    if (err) break;
                    it = todo.begin();
                    while (it != todo.end() && it->state != 0) ++it;
                    if (it == todo.end()) break;
                    it->state = 1;
                }
                bool split = false;
                double over = 0;
                try {
                    search_impl(ix, it->part, &it->res, ctl);
                } catch (const PartTooLarge &e) {
                    split = true;
                    over = e.over;
                    if (getenv("LM_DEBUG_MEM")) fprintf(stderr, "[lm] lane %d: a part of %d queries is halved: its seed anchors exceed the lane's share of the budget\n", lane, it->part->nq);
                } catch (const DeviceOOM &e) {
                    if (it->part->nq < 2) throw;
                    if (getenv("LM_DEBUG_MEM")) fprintf(stderr, "[lm] lane %d: a part of %d queries is halved after: %s\n", lane, it->part->nq, e.what());
                    drop_scratch(ix, e.what());
                    split = true;
                }
                if (split) {
                    std::vector<lm_qbatch *> pieces;
                    if (over > 0) {
                        pieces = split_qbatch(ix, it->part, over);
                    } else {
                        auto ab = halve_qbatch(ix, it->part);
                        pieces = {ab.first, ab.second};
                    }
                    std::lock_guard<std::mutex> l(lm_);
                    delete it->part;
                    it->part = pieces[0]; // this entry becomes the first piece, the others follow it in order
                    it->state = 0;
                    for (auto *str : it->res.strings) delete str;
                    it->res.strings.clear();
                    it->res.rows.clear();
                    auto nx = std::next(it);
                    for (size_t pi = 1; pi < pieces.size(); pi++) {
                        auto ins = todo.emplace(nx);
                        ins->part = pieces[pi];
                        ins->state = 0;
                    }
                } else {
                    std::lock_guard<std::mutex> l(lm_);
                    it->state = 2;
                }
            }
        } catch (...) {
            std::lock_guard<std::mutex> l(lm_);
            if (!err) err = std::current_exception();
        }
        tls_stream = saved_stream;
        tls_tmp = saved_tmp;
        tls_lane = 0;
    };
    std::thread second;
    if (lanes == 2) second = std::thread(lane_fn, 1);
    lane_fn(0);
    if (second.joinable()) second.join();
    if (getenv("LM_DEBUG_MEM")) { // what the scratch of this search took from the device
        size_t fr = 0, tot = 0;
        (void)hipMemGetInfo(&fr, &tot);
        fprintf(stderr, "[lm] scratch after a search of %zu parts on %d lane(s) (budget / %d): lane slabs %.2f GB; overflow slabs: lane 0 %.2f GB (%lld device allocations so far), lane 1 %.2f GB (%lld); device free %.2f GB; all buffers of this library %.2f GB\n",
                todo.size(), lanes, blanes, (double)ix->lane_slabs.bytes() / 1e9, (double)ix->arena[0].slab_bytes / 1e9, (long long)ix->arena[0].slab_allocs,
                (double)ix->arena[1].slab_bytes / 1e9, (long long)ix->arena[1].slab_allocs, (double)fr / 1e9, (double)g_dbuf_bytes.load() / 1e9);
    }
    ix->active_lanes = 1;
    ix->budget_lanes = 1;
    // the (possibly finer) split stays with the batch handle
    qb->parts.clear();
    for (auto &ps : todo) qb->parts.push_back(ps.part);
    if (err) std::rethrow_exception(err);
    memset(&res->stats, 0, sizeof res->stats);
    for (auto &ps : todo) {
        lm_result &pr = ps.res;
        for (auto &r : pr.rows) r.query += ps.part->q0;
        res->rows.insert(res->rows.end(), pr.rows.begin(), pr.rows.end());
        res->strings.insert(res->strings.end(), pr.strings.begin(), pr.strings.end());
        pr.strings.clear();
        const int64_t *a = &pr.stats.query_bases;
        int64_t *b = &res->stats.query_bases;
        for (int i = 0; i < 14; i++) b[i] += a[i]; // the int64 counters of lm_stage_stats
        const double *c = &pr.stats.ms_mask;
        double *d = &res->stats.ms_mask;
        for (int i = 0; i < 9; i++) d[i] += c[i];
    }
}

lm_status lm_search_resident(lm_index *ix, lm_qbatch *qb, lm_result **out) {
    *out = nullptr;
    if (!ix || !qb) return LM_ERR_ARG;
    lm_result *res = new lm_result();
    try {
        search_parts(ix, qb, res, nullptr);
    } catch (const std::exception &e) {
        ix->err = e.what();
        delete res;
        return LM_ERR_HIP;
    }
    *out = res;
    return LM_OK;
}

lm_status lm_search_scores(lm_index *ix, lm_qbatch *qb, lm_stage **out, size_t *n, const uint32_t **query,
                           const uint64_t **batch_genome, const float **score) {
    *out = nullptr;
    if (!ix || !qb || !n) return LM_ERR_ARG;
    lm_stage *sg = new lm_stage();
    try {
        lm_result tmp;
        SearchCtl ctl;
        ctl.sc_query = &sg->u32a;
        ctl.sc_bg = &sg->u64a;
        ctl.sc_score = &sg->f32a;
        search_parts(ix, qb, &tmp, &ctl);
    } catch (const std::exception &e) {
        ix->err = e.what();
        delete sg;
        return LM_ERR_HIP;
    }
    *n = sg->u32a.size();
    *query = sg->u32a.data();
    *batch_genome = sg->u64a.data();
    *score = sg->f32a.data();
    *out = sg;
    return LM_OK;
}

lm_status lm_search_resident_keep(lm_index *ix, lm_qbatch *qb, const uint32_t *keep_query, const uint64_t *keep_bg, size_t nkeep,
                                  lm_result **out) {
    *out = nullptr;
    if (!ix || !qb || (nkeep && (!keep_query || !keep_bg))) return LM_ERR_ARG;
    lm_result *res = new lm_result();
    try {
        std::unordered_map<uint64_t, std::vector<uint32_t>> keep;
        for (size_t i = 0; i < nkeep; i++) keep[keep_bg[i]].push_back(keep_query[i]);
        for (auto &kv : keep) std::sort(kv.second.begin(), kv.second.end());
        SearchCtl ctl;
        ctl.keep = &keep;
        search_parts(ix, qb, res, &ctl);
    } catch (const std::exception &e) {
        ix->err = e.what();
        delete res;
        return LM_ERR_HIP;
    }
    *out = res;
    return LM_OK;
}

lm_status lm_index_set_genome_filter(lm_index *ix, const uint64_t *keys, size_t n) {
    if (!ix || (n && !keys)) return LM_ERR_ARG;
    try {
        std::lock_guard<std::mutex> lock(ix->mu);
        HIPCHK(hipSetDevice(ix->device));
        if (n == 0) {
            ix->view.g_keep = nullptr;
            return LM_OK;
        }
        std::vector<uint32_t> bits((ix->host.genomes.size() + 31) / 32 + 1, 0);
        for (size_t i = 0; i < n; i++) {
            auto it = ix->bg2local.find(keys[i]); // genomes of other shards are not ours to filter
            if (it != ix->bg2local.end()) bits[(size_t)it->second >> 5] |= 1u << (it->second & 31);
        }
        h2d(ix, ix->d_g_keep, bits);
        sync(ix);
        ix->view.g_keep = ix->d_g_keep.p;
        return LM_OK;
    } catch (const std::exception &e) {
        ix->err = e.what();
        return LM_ERR_HIP;
    }
}

lm_status lm_search_batch(lm_index *ix, const lm_query *queries, size_t nq, lm_result **out) {
    *out = nullptr;
    lm_qbatch *qb = nullptr;
    lm_status s = lm_qbatch_upload(ix, queries, nq, &qb);
    if (s != LM_OK) return s;
    s = lm_search_resident(ix, qb, out);
    lm_qbatch_free(qb);
    return s;
}

size_t lm_result_rows(const lm_result *res, const lm_hsp **rows) {
    *rows = res->rows.data();
    return res->rows.size();
}
void lm_result_stats(const lm_result *res, lm_stage_stats *stats) { *stats = res->stats; }
void lm_result_free(lm_result *res) { delete res; }

int lm_format_row_ex(const lm_hsp *h, const char *query_id, uint32_t qlen, int flags, char *buf, size_t buflen) {
    // search.go:483-520: the two Fprintf forms differ in the sseqid column only
    char sseq[64];
    std::string sseq_long;
    const char *sseqid = h->seq_id ? h->seq_id : "";
    if (flags & LM_ROW_SSEQ_IDX) {
        int m = snprintf(sseq, sizeof sseq, "c%d/%d:s%d/%d:", h->chunk_idx + 1, h->nchunks, h->seq_idx + 1, h->nseqs);
        (void)m;
        sseq_long = std::string(sseq) + sseqid;
        sseqid = sseq_long.c_str();
    }
    int n = snprintf(buf, buflen, "%s\t%u\t%u\t%s\t%s\t%.3f\t%d\t%d\t%.3f\t%d\t%.3f\t%d\t%d\t%d\t%d\t%d\t%c\t%d\t%.2e\t%d",
                     query_id, qlen, h->hits, h->genome_id, sseqid, h->qcov_genome, h->cls, h->hsp, h->qcov_hsp,
                     h->aligned_length, h->pident, h->gaps, h->qbegin + 1, h->qend + 1, h->tbegin + 1, h->tend + 1,
                     h->rc ? '-' : '+', h->seq_len, h->evalue, h->bitscore);
    if ((flags & LM_ROW_ALL) && n > 0) {
        int m = snprintf((size_t)n < buflen ? buf + n : nullptr, (size_t)n < buflen ? buflen - n : 0, "\t%s\t%s\t%s\t%s",
                         h->cigar ? h->cigar : "", h->qseq ? h->qseq : "", h->sseq ? h->sseq : "",
                         h->align ? h->align : "");
        n += m;
    }
    return n;
}
int lm_format_row(const lm_hsp *h, const char *query_id, uint32_t qlen, int more_columns, char *buf, size_t buflen) {
    return lm_format_row_ex(h, query_id, qlen, more_columns ? LM_ROW_ALL : 0, buf, buflen);
}
// The printer's loop over a whole batch (search.go:468-523 is one writer goroutine; a C3 step emits 5 M rows = 0.7 GB of text):
// every row formatted as by lm_format_row_ex + a newline, by the host threads into ONE buffer, in row order.
lm_status lm_format_rows(const lm_hsp *rows, size_t n, const char *const *query_ids, const uint32_t *query_lens, size_t nq, int flags, char **text,
                         size_t *len) {
    if (!text || !len || (n > 0 && (!rows || !query_ids || !query_lens))) return LM_ERR_ARG;
    *text = nullptr;
    *len = 0;
    for (size_t i = 0; i < n; i++)
        if (rows[i].query >= nq) return LM_ERR_ARG;
    const int64_t grain = 8192;
    const int64_t nparts = ((int64_t)n + grain - 1) / grain;
    std::vector<std::string> parts((size_t)nparts);
    try {
        parallel_for(nparts, 1, [&](int64_t p0, int64_t p1) {
            std::vector<char> line((size_t)1 << 16);
            for (int64_t p = p0; p < p1; p++) {
                std::string &out = parts[(size_t)p];
                const size_t b = (size_t)(p * grain), e = std::min(n, (size_t)((p + 1) * grain));
                out.reserve((e - b) * 192);
                for (size_t i = b; i < e; i++) {
                    const uint32_t q = rows[i].query;
                    int need = lm_format_row_ex(&rows[i], query_ids[q] ? query_ids[q] : "", query_lens[q], flags, line.data(), line.size());
                    if (need < 0) need = 0;
                    if ((size_t)need >= line.size()) { // (-a rows carry the aligned sequences: as long as the HSP)
                        line.resize((size_t)need + 1);
                        need = lm_format_row_ex(&rows[i], query_ids[q] ? query_ids[q] : "", query_lens[q], flags, line.data(), line.size());
                    }
                    out.append(line.data(), (size_t)need);
                    out.push_back('\n');
                }
            }
        });
        std::vector<size_t> off((size_t)nparts + 1, 0);
        for (int64_t p = 0; p < nparts; p++) off[(size_t)p + 1] = off[(size_t)p] + parts[(size_t)p].size();
        char *buf = (char *)malloc(off[(size_t)nparts] + 1);
        if (!buf) return LM_ERR_NOMEM;
        parallel_for(nparts, 1, [&](int64_t p0, int64_t p1) {
            for (int64_t p = p0; p < p1; p++) memcpy(buf + off[(size_t)p], parts[(size_t)p].data(), parts[(size_t)p].size());
        });
        buf[off[(size_t)nparts]] = 0;
        *text = buf;
        *len = off[(size_t)nparts];
    } catch (const std::exception &) {
        return LM_ERR_NOMEM;
    }
    return LM_OK;
}
const char *lm_tsv_header(int more_columns) {
    return more_columns ? "query\tqlen\thits\tsgenome\tsseqid\tqcovGnm\tcls\thsp\tqcovHSP\talenHSP\tpident\tgaps\tqstart\tqend\tsstart\tsend\tsstr\tslen\tevalue\tbitscore\tcigar\tqseq\tsseq\talign"
                        : "query\tqlen\thits\tsgenome\tsseqid\tqcovGnm\tcls\thsp\tqcovHSP\talenHSP\tpident\tgaps\tqstart\tqend\tsstart\tsend\tsstr\tslen\tevalue\tbitscore";
}

void lm_stage_free(lm_stage *s) { delete s; }

// ---- stage-level entry points ----------------------------------------------------------------------------------
lm_status lm_mask_batch(lm_index *ix, const lm_query *queries, size_t nq, lm_stage **out, const uint64_t **kmers,
                        const int64_t **loc_off, const int32_t **locs) {
    *out = nullptr;
    lm_qbatch *qb = nullptr;
    lm_status s = lm_qbatch_upload(ix, queries, nq, &qb);
    if (s != LM_OK) return s;
    lm_stage *sg = new lm_stage();
    try {
        Work w(ix, qb);
        stage_kmers(w);
        stage_mask(w);
        int64_t nqm = (int64_t)nq * ix->host.M;
        std::vector<int64_t> lo, hi;
        std::vector<uint32_t> vals;
        d2h(ix, sg->u64a, w.kmers.p, (size_t)nqm);
        d2h(ix, lo, w.klo.p, (size_t)nqm);
        d2h(ix, hi, w.khi.p, (size_t)nqm);
        d2h(ix, vals, (const uint32_t *)w.v_all, (size_t)(2 * qb->total_pos));
        sync(ix);
        sg->i64a.assign(nqm + 1, 0);
        for (int64_t i = 0; i < nqm; i++) sg->i64a[i + 1] = sg->i64a[i] + (hi[i] - lo[i]);
        sg->i32a.resize((size_t)sg->i64a[nqm]);
        for (int64_t i = 0; i < nqm; i++)
            for (int64_t j = lo[i]; j < hi[i]; j++) sg->i32a[sg->i64a[i] + (j - lo[i])] = (int32_t)vals[j];
    } catch (const std::exception &e) {
        ix->err = e.what();
        delete sg;
        lm_qbatch_free(qb);
        return LM_ERR_HIP;
    }
    lm_qbatch_free(qb);
    *kmers = sg->u64a.data();
    *loc_off = sg->i64a.data();
    *locs = sg->i32a.data();
    *out = sg;
    return LM_OK;
}

lm_status lm_seed_chain_batch(lm_index *ix, const lm_query *queries, size_t nq, lm_stage **out, size_t *npairs,
                              const lm_pair **pairs, const lm_anchor **raw, const lm_anchor **cleared,
                              const int64_t **chain_ptr, const int32_t **chain_idx) {
    *out = nullptr;
    *npairs = 0;
    lm_qbatch *qb = nullptr;
    lm_status s = lm_qbatch_upload(ix, queries, nq, &qb);
    if (s != LM_OK) return s;
    lm_stage *sg = new lm_stage();
    try {
        lm_stage_stats st;
        memset(&st, 0, sizeof st);
        Work w(ix, qb);
        stage_kmers(w);
        stage_mask(w);
        stage_lookup(w, st);
        sg->i64a.assign(1, 0);
        if (w.nseg > 0) {
            stage_chain1(w);
            int nseg = w.nseg;
            int64_t T = w.n_anchors;
            std::vector<uint64_t> segA, B;
            std::vector<int64_t> seg_off;
            std::vector<int32_t> seg_n, seg_nch, coff, cidx;
            std::vector<float> score;
            std::vector<LmSub> subs;
            d2h(ix, segA, w.segA.p, (size_t)nseg);
            d2h(ix, B, w.B0.p, (size_t)T);
            d2h(ix, seg_off, w.seg_off.p, (size_t)nseg + 1);
            d2h(ix, seg_n, w.seg_n.p, (size_t)nseg);
            d2h(ix, seg_nch, w.seg_nch.p, (size_t)nseg);
            d2h(ix, score, w.seg_score.p, (size_t)nseg);
            d2h(ix, subs, w.subs.p, (size_t)T);
            d2h(ix, coff, w.chain_off_pool.p, (size_t)T + 4 * (size_t)nseg);
            d2h(ix, cidx, w.chain_idx_pool.p, 2 * (size_t)T + 8 * (size_t)nseg);
            sync(ix);
            sg->raw.resize((size_t)T);
            for (int64_t i = 0; i < T; i++) {
                LmSub x = lm_unpack_anchor(B[i]);
                sg->raw[i] = {x.qbegin, x.tbegin, x.len, x.trc, x.qrc, 0};
            }
            for (int sidx = 0; sidx < nseg; sidx++) {
                lm_pair p;
                p.query = (uint32_t)(segA[sidx] >> 34);
                p.batch_genome = segA[sidx] & ((1ull << 34) - 1);
                p.raw_off = seg_off[sidx];
                p.raw_n = seg_off[sidx + 1] - seg_off[sidx];
                p.clr_off = (int64_t)sg->cleared.size();
                p.clr_n = seg_n[sidx];
                for (int i = 0; i < seg_n[sidx]; i++) {
                    const LmSub &x = subs[seg_off[sidx] + i];
                    sg->cleared.push_back({x.qbegin, x.tbegin, x.len, x.trc, x.qrc, 0});
                }
                p.score = score[sidx];
                p.chain_off = (int64_t)sg->i64a.size() - 1;
                p.chain_n = seg_nch[sidx];
                const int32_t *co = coff.data() + seg_off[sidx] + 4ll * sidx;
                const int32_t *ci = cidx.data() + 2 * seg_off[sidx] + 8ll * sidx;
                for (int c = 0; c < seg_nch[sidx]; c++) {
                    for (int j = co[c]; j < co[c + 1]; j++) sg->i32a.push_back(ci[j]);
                    sg->i64a.push_back((int64_t)sg->i32a.size());
                }
                sg->pairs.push_back(p);
            }
        }
    } catch (const std::exception &e) {
        ix->err = e.what();
        delete sg;
        lm_qbatch_free(qb);
        return LM_ERR_HIP;
    }
    lm_qbatch_free(qb);
    *npairs = sg->pairs.size();
    *pairs = sg->pairs.data();
    *raw = sg->raw.data();
    *cleared = sg->cleared.data();
    *chain_ptr = sg->i64a.data();
    *chain_idx = sg->i32a.data();
    *out = sg;
    return LM_OK;
}

lm_status lm_pseudoalign_batch(lm_index *ix, const lm_query *queries, size_t nq, const lm_query *targets,
                               const uint32_t *qidx, const uint32_t *qbegin, const uint32_t *qend, size_t nproblems,
                               lm_stage **out, const int64_t **res_off, const lm_chain2 **resv) {
    *out = nullptr;
    lm_qbatch *qb = nullptr;
    lm_status s = lm_qbatch_upload(ix, queries, nq, &qb);
    if (s != LM_OK) return s;
    lm_stage *sg = new lm_stage();
    try {
        lm_stage_stats st;
        memset(&st, 0, sizeof st);
        Work w(ix, qb);
        stage_kmers(w);
        AlignCtx a;
        a.ix = ix;
        a.qb = qb;
        a.w = &w;
        a.stats = &st;
        // explicit windows: upload targets as the window buffer, build tasks on the host
        std::vector<Task> ht(nproblems);
        int64_t off = 0;
        for (size_t i = 0; i < nproblems; i++) {
            Task t;
            memset(&t, 0, sizeof t);
            t.seg = (uint32_t)i;
            t.q = qidx[i];
            t.g = -1; // windows come from the caller, k_extract_windows skips them
            t.bg = 0;
            t.qBegin = (int32_t)qbegin[i];
            t.qEnd = (int32_t)qend[i];
            t.wlen = (int32_t)targets[i].len;
            t.woff = off;
            off += t.wlen;
            ht[i] = t;
        }
        std::vector<uint8_t> wb((size_t)off + 64, 'A');
        for (size_t i = 0; i < nproblems; i++)
            if (targets[i].len) memcpy(&wb[(size_t)ht[i].woff], targets[i].seq, targets[i].len);
        a.wbuf.ensure(wb.size());
        HIPCHK(hipMemcpyAsync(a.wbuf.p, wb.data(), wb.size(), hipMemcpyHostToDevice, S(ix)));
        std::vector<int64_t> ro;
        std::vector<LmChain2> rv;
        TaskSpan sp;
        sp.p = ht.data();
        sp.n = ht.size();
        run_pseudo(a, sp, ro, rv);
        sg->i64a = ro;
        sg->chains2.resize(rv.size());
        for (size_t i = 0; i < rv.size(); i++)
            sg->chains2[i] = {rv[i].qbegin, rv[i].qend, rv[i].tbegin, rv[i].tend, rv[i].nanchors, rv[i].matched_bases,
                              rv[i].aligned_bases_q, rv[i].aligned_bases_t, rv[i].pident};
    } catch (const std::exception &e) {
        ix->err = e.what();
        delete sg;
        lm_qbatch_free(qb);
        return LM_ERR_HIP;
    }
    lm_qbatch_free(qb);
    *res_off = sg->i64a.data();
    *resv = sg->chains2.data();
    *out = sg;
    return LM_OK;
}

lm_status lm_wfa_batch(lm_index *ix, const lm_query *q, const lm_query *t, size_t n, lm_stage **out, const lm_wfa **resv,
                       const uint64_t **ops) {
    *out = nullptr;
    if (!ix) return LM_ERR_ARG;
    lm_stage *sg = new lm_stage();
    try {
        HIPCHK(hipSetDevice(ix->device));
        lm_stage_stats st;
        memset(&st, 0, sizeof st);
        AlignCtx a;
        a.ix = ix;
        a.qb = nullptr;
        a.w = nullptr;
        a.stats = &st;
        std::vector<uint8_t> buf;
        std::vector<int64_t> qo(n), to(n);
        for (size_t i = 0; i < n; i++) {
            qo[i] = (int64_t)buf.size();
            buf.insert(buf.end(), q[i].seq, q[i].seq + q[i].len);
            to[i] = (int64_t)buf.size();
            buf.insert(buf.end(), t[i].seq, t[i].seq + t[i].len);
        }
        buf.resize(buf.size() + 64, 'A');
        a.wbuf.ensure(buf.size());
        HIPCHK(hipMemcpyAsync(a.wbuf.p, buf.data(), buf.size(), hipMemcpyHostToDevice, S(ix)));
        std::vector<WfaIn> win(n);
        for (size_t i = 0; i < n; i++) {
            win[i].q = a.wbuf.p + qo[i];
            win[i].t = a.wbuf.p + to[i];
            win[i].qlen = (int32_t)q[i].len;
            win[i].tlen = (int32_t)t[i].len;
        }
        std::vector<WfaOut> wout;
        std::vector<int64_t> ops_off;
        run_wfa(a, win, wout, sg->u64a, ops_off, true);
        sg->wfa.resize(n);
        for (size_t i = 0; i < n; i++) {
            const LmWfaOut &r = wout[i].r;
            sg->wfa[i] = {r.status, r.score, r.qbegin, r.qend, r.tbegin, r.tend, r.align_len, r.matches, r.gaps,
                          r.gap_regions, ops_off[i], r.nops};
        }
    } catch (const std::exception &e) {
        ix->err = e.what();
        delete sg;
        return LM_ERR_HIP;
    }
    *resv = sg->wfa.data();
    *ops = sg->u64a.data();
    *out = sg;
    return LM_OK;
}


// genome_id / seq_id of a row that came from another process: every shard holds the names of all genomes (this shard's in
// host.genomes, the others' in host.others); the names of a synthetic set are a function of the genome number, made once per
// genome and kept with the handle
static void attach_name(lm_index *ix, lm_hsp &h) {
    const HostIndex &H = ix->host;
    auto it = ix->bg2local.find(h.batch_genome);
    const HostGenome *G = nullptr;
    if (it != ix->bg2local.end()) {
        G = &H.genomes[it->second];
    } else {
        auto io = H.other_of.find(h.batch_genome);
        if (io != H.other_of.end()) G = &H.others[io->second];
    }
    if (G) {
        h.genome_id = G->id.c_str();
        if (h.seq_idx >= 0 && h.seq_idx < (int)G->seq_ids.size()) h.seq_id = G->seq_ids[h.seq_idx].c_str();
        return;
    }
    if (!H.synthetic) return;
    const long long g = (long long)((h.batch_genome >> 17) * 5000 + (h.batch_genome & 0x1ffff));
    auto make = [&](const char **a, const char **b) { // (under the unique lock)
        char nm[64];
        snprintf(nm, sizeof nm, "SYN_%09lld.1", g);
        ix->syn_store.emplace_back(nm);
        *a = ix->syn_store.back().c_str();
        snprintf(nm, sizeof nm, "syn%09lld_c1", g);
        ix->syn_store.emplace_back(nm);
        *b = ix->syn_store.back().c_str();
    };
    if (g >= 0 && g < H.synth_genomes && ix->syn_dense) { // one load per name once a genome has been seen (4.7 M rows per C3 step)
        std::atomic<const char *> *e = ix->syn_dense.get() + 2 * g;
        const char *b = e[1].load(std::memory_order_acquire); // (the sequence id is published last)
        if (!b) {
            std::unique_lock<std::shared_mutex> wl(ix->syn_mu);
            b = e[1].load(std::memory_order_relaxed);
            if (!b) {
                const char *a = nullptr;
                make(&a, &b);
                e[0].store(a, std::memory_order_relaxed);
                e[1].store(b, std::memory_order_release);
            }
        }
        h.genome_id = e[0].load(std::memory_order_relaxed);
        h.seq_id = b;
        return;
    }
    {
        std::shared_lock<std::shared_mutex> rl(ix->syn_mu);
        auto is = ix->syn_names.find(h.batch_genome);
        if (is != ix->syn_names.end()) {
            h.genome_id = is->second.first;
            h.seq_id = is->second.second;
            return;
        }
    }
    std::unique_lock<std::shared_mutex> wl(ix->syn_mu);
    auto is = ix->syn_names.find(h.batch_genome);
    if (is == ix->syn_names.end()) {
        const char *a = nullptr, *b = nullptr;
        make(&a, &b);
        is = ix->syn_names.emplace(h.batch_genome, std::make_pair(a, b)).first;
    }
    h.genome_id = is->second.first;
    h.seq_id = is->second.second;
}
// the handle's scratch slabs lent to the shard merge between two searches (lm_merge.h)
void lm_scratch_session_begin(lm_index *ix) {
    if (!ix) return;
    ix->mu.lock();
    (void)hipSetDevice(ix->device);
}
void lm_scratch_session_end(lm_index *ix) {
    if (ix) ix->mu.unlock();
}
void *lm_scratch_borrow(lm_index *ix, size_t bytes) {
    if (!ix || bytes == 0) return nullptr;
    try {
        return ix->arena[0].alloc(bytes);
    } catch (const std::exception &e) {
        ix->err = e.what();
        return nullptr;
    }
}
void lm_scratch_return(lm_index *ix, void *p) {
    if (ix && p) (void)ix->arena[0].release(p);
}
void lm_attach_names(lm_index *ix, lm_hsp *rows, size_t n) {
    if (!ix || n == 0) return;
    if (ix->host.synthetic && !ix->syn_dense && ix->host.synth_genomes > 0) {
        std::unique_lock<std::shared_mutex> wl(ix->syn_mu);
        if (!ix->syn_dense) {
            const size_t m = 2 * (size_t)ix->host.synth_genomes;
            ix->syn_dense.reset(new std::atomic<const char *>[m]);
            for (size_t i = 0; i < m; i++) ix->syn_dense[i].store(nullptr, std::memory_order_relaxed);
        }
    }
    parallel_for((int64_t)n, 4096, [&](int64_t a, int64_t b) {
        for (int64_t i = a; i < b; i++) {
            if (i > a && rows[i].batch_genome == rows[i - 1].batch_genome && rows[i].seq_idx == rows[i - 1].seq_idx) { // a genome's rows are together
                rows[i].genome_id = rows[i - 1].genome_id;
                rows[i].seq_id = rows[i - 1].seq_id;
                continue;
            }
            attach_name(ix, rows[i]);
        }
    });
}

// ---- merging the rows of genome shards ------------------------------------------------------------
// Host-only: no device work, callable without a GPU (idx may be NULL: names are then left NULL).
lm_status lm_merge_sharded(lm_index *ix, const lm_hsp *const *rows, const size_t *nrows, int nshards, lm_result **out) {
    *out = nullptr;
    if (nshards < 1 || !rows || !nrows) return LM_ERR_ARG;
    lm_result *res = new lm_result();
    try {
        memset(&res->stats, 0, sizeof res->stats);
        size_t total = 0;
        for (int r = 0; r < nshards; r++) total += nrows[r];
        // Per shard the rows are grouped by query, ascending.  First the (query -> row range) table of every shard and the
        // output position of every query; then the queries are merged independently on the host threads (at 8 shards x
        // 6e5 rows the single-threaded merge was 0.33 s of a 2.1-s step: the serial part of the N-GPU run).
        struct QRange {
            uint32_t q;
            size_t b, e;
        };
        std::vector<std::vector<QRange>> qr((size_t)nshards);
        for (int r = 0; r < nshards; r++)
            for (size_t i = 0; i < nrows[r];) {
                size_t e = i + 1;
                while (e < nrows[r] && rows[r][e].query == rows[r][i].query) e++;
                if (!qr[r].empty() && rows[r][i].query <= qr[r].back().q) throw std::runtime_error("lm_merge_sharded: rows of a shard are not grouped by ascending query");
                qr[r].push_back(QRange{rows[r][i].query, i, e});
                i = e;
            }
        struct QJob {
            uint32_t q;
            size_t out;
            std::vector<std::pair<int, int>> parts; // (shard, index into qr[shard])
        };
        std::vector<QJob> jobs;
        {
            std::vector<size_t> at((size_t)nshards, 0);
            size_t out = 0;
            while (true) {
                uint32_t q = 0;
                bool any = false;
                for (int r = 0; r < nshards; r++)
                    if (at[r] < qr[r].size() && (!any || qr[r][at[r]].q < q)) {
                        q = qr[r][at[r]].q;
                        any = true;
                    }
                if (!any) break;
                QJob j;
                j.q = q;
                j.out = out;
                for (int r = 0; r < nshards; r++)
                    if (at[r] < qr[r].size() && qr[r][at[r]].q == q) {
                        j.parts.emplace_back(r, (int)at[r]);
                        out += qr[r][at[r]].e - qr[r][at[r]].b;
                        at[r]++;
                    }
                jobs.push_back(std::move(j));
            }
        }
        res->rows.resize(total);
        parallel_for((int64_t)jobs.size(), 16, [&](int64_t j0, int64_t j1) {
            struct Grp {
                int rank;
                size_t b, e;
                uint64_t bg;
                double best;
            };
            std::vector<Grp> grps;
            for (int64_t ji = j0; ji < j1; ji++) {
                const QJob &job = jobs[(size_t)ji];
                grps.clear();
                for (auto &pr : job.parts) {
                    const int r = pr.first;
                    const QRange &R = qr[r][(size_t)pr.second];
                    for (size_t i = R.b; i < R.e;) {
                        Grp g{r, i, i, rows[r][i].batch_genome, 0.0};
                        while (g.e < R.e && rows[r][g.e].batch_genome == g.bg) {
                            const double sim = (double)rows[r][g.e].bitscore * rows[r][g.e].pident; // SimilarityScore (:2352,2621)
                            if (sim > g.best) g.best = sim;
                            g.e++;
                        }
                        grps.push_back(g);
                        i = g.e;
                    }
                }
                // genomes by the similarity of their best HSP cluster, descending (lib-index-search.go:2919-2921), ties by key
                std::stable_sort(grps.begin(), grps.end(), [](const Grp &x, const Grp &y) {
                    if (x.best != y.best) return x.best > y.best;
                    return x.bg < y.bg;
                });
                size_t w = job.out;
                for (const Grp &g : grps)
                    for (size_t i = g.b; i < g.e; i++) {
                        lm_hsp h = rows[g.rank][i];
                        h.hits = (uint32_t)grps.size(); // search.go:463,494: subject genomes of the query, over all shards
                        h.genome_id = h.seq_id = nullptr;
                        h.cigar = h.qseq = h.sseq = h.align = nullptr; // process-local addresses of another rank
                        if (ix) attach_name(ix, h);
                        res->rows[w++] = h;
                    }
            }
        });
        res->stats.rows = (int64_t)res->rows.size();
    } catch (const std::exception &e) {
        if (ix) ix->err = e.what();
        delete res;
        return LM_ERR_NOMEM;
    }
    *out = res;
    return LM_OK;
}

// -n/--top-n-genomes over genome shards: the cut of lib-index-search.go:1781-1805 needs the chaining scores of ALL shards.
// Every shard reports its candidates (lm_search_scores: per query its own top-N (query, genome, score)), the host gathers
// them, lm_topn_merge picks the global top-N per query by (score desc, genome key asc), and every shard then searches
// with that keep list (lm_search_resident_keep).
lm_status lm_topn_merge(int nshards, const uint32_t *const *query, const uint64_t *const *bg, const float *const *score,
                        const size_t *n, int top_n, uint32_t **out_query, uint64_t **out_bg, size_t *out_n) {
    if (nshards < 1 || top_n < 1 || !out_query || !out_bg || !out_n) return LM_ERR_ARG;
    struct C {
        uint32_t q;
        uint64_t bg;
        float s;
    };
    std::vector<C> all;
    for (int r = 0; r < nshards; r++)
        for (size_t i = 0; i < n[r]; i++) all.push_back({query[r][i], bg[r][i], score[r][i]});
    std::stable_sort(all.begin(), all.end(), [](const C &x, const C &y) {
        if (x.q != y.q) return x.q < y.q;
        if (x.s != y.s) return x.s > y.s;
        return x.bg < y.bg;
    });
    std::vector<C> keep;
    for (size_t i = 0; i < all.size();) {
        size_t e = i;
        while (e < all.size() && all[e].q == all[i].q) e++;
        for (size_t j = i; j < e && j < i + (size_t)top_n; j++) keep.push_back(all[j]);
        i = e;
    }
    *out_n = keep.size();
    *out_query = (uint32_t *)malloc(sizeof(uint32_t) * std::max<size_t>(keep.size(), 1));
    *out_bg = (uint64_t *)malloc(sizeof(uint64_t) * std::max<size_t>(keep.size(), 1));
    if (!*out_query || !*out_bg) return LM_ERR_NOMEM;
    for (size_t i = 0; i < keep.size(); i++) {
        (*out_query)[i] = keep[i].q;
        (*out_bg)[i] = keep[i].bg;
    }
    return LM_OK;
}
void lm_free(void *p) { free(p); }

} // extern "C"
