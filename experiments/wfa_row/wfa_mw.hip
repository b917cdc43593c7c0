// wfa_mw.hip - EXPERIMENT (staged for the next round; not linked into liblexicmap_hip.so): the WFA kernel with a workgroup of
// four wavefronts per alignment (wfa_mw_fwd.h) beside the product's k_wfa_lean<8 / 16> on the same long problems: results
// compared struct by struct and operation by operation, both timed with HIP events.  One translation unit with the product's
// kernels (included as text: bt_walk / bt_replay / launch_wfa reused unchanged); builds into libwfa_mw_exp.so.
#include "../../lexicmap_amd/csrc/lm_kernels.hip"

#include <stdio.h>
#include <string.h>

#include <vector>

namespace lm {

#define WR_DEV __device__ __forceinline__
#define WR_TID ((int)threadIdx.x)
#define WR_BALLOT(p) __ballot(p)
#define WR_BARRIER() __syncthreads()
#define WR_UNIFORM(x) __builtin_amdgcn_readfirstlane((int)(x))
#define WR_CLZ(x) __clz((int)(x))
#define WR_CLZLL(x) __clzll((long long)(x))
#define wr_pk_min_u16 pk_min_u16
#define WR_WAVE_MIN_I32(v) wave_min_i32(v)
#define WR_WAVE_PKMIN_U16(v) wave_pkmin_u16(v)
#define WR_NULL_OFF LM_NULL_OFF

#include "wfa_mw_fwd.h"

// Persistent workgroups of four wavefronts; each pops ONE problem at a time: forward pass by all four, then the backtrace by
// the first wavefront (bt_walk / bt_replay of k_wfa_lean, unchanged) while the others wait at the barrier.
template <int NCW, bool WIN>
__global__ __launch_bounds__(MW_THREADS) void k_wfa_mw(const WfaIn *__restrict__ in, int64_t n, const int32_t *__restrict__ todo, int64_t ntodo,
                                                        int32_t *__restrict__ hdr_pool, int64_t hdr_stride, uint8_t *__restrict__ arena_pool,
                                                        int64_t arena_stride, uint64_t *__restrict__ ops_pool, unsigned int *__restrict__ queue,
                                                        int seq_words, int want_ops, WfaOut *__restrict__ out) {
    constexpr int W = MW_THREADS * NCW;
    constexpr int RING_BYTES = 9 * W * 4 > (int)sizeof(BtLds) ? 9 * W * 4 : (int)sizeof(BtLds);
    __shared__ __attribute__((aligned(16))) uint8_t ring_raw[RING_BYTES]; // the backtrace walk reuses the ring (dead by then)
    __shared__ int32_t red[40];
    __shared__ unsigned int sh_x;
    // WIN: the two sequence windows; otherwise both whole packed sequences in dynamic LDS (seq_words + 2 words each)
    __shared__ uint32_t qwin_buf[WIN ? MW_WINW + 2 : 1], twin_buf[WIN ? MW_WINW + 2 : 1];
    extern __shared__ uint32_t seq_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    int32_t *hdr2 = hdr_pool + (int64_t)blockIdx.x * hdr_stride;
    uint8_t *bt = arena_pool + (int64_t)blockIdx.x * arena_stride;
    const int max_score = (int)(hdr_stride / 2 - 2) * 2;
    if (tid == 0) sh_x = atomicAdd(queue, 1u);
    while (true) {
        __syncthreads();
        const unsigned int x = (unsigned int)__builtin_amdgcn_readfirstlane((int)sh_x);
        __syncthreads();
        if ((int64_t)x >= ntodo) break;
        const int64_t i = todo ? todo[x] : (int64_t)x;
        if (i < 0 || i >= n) break;
        const WfaIn w = in[i];
        MwProb p;
        p.q = w.q;
        p.t = w.t;
        p.plen = w.qlen;
        p.tlen = w.tlen;
        p.hdr2 = hdr2;
        p.bt = bt;
        p.arena_cap = (int32_t)std::min<int64_t>(arena_stride - 16, 2000000000);
        p.max_score = max_score;
        MwLds L;
        L.ring = (int32_t *)ring_raw;
        L.qbuf = WIN ? qwin_buf : seq_lds;
        L.tbuf = WIN ? twin_buf : seq_lds + seq_words + 2;
        L.red = red;
        MwRes res;
        wfa_mw_forward<NCW, WIN>(p, L, seq_words, &res);
        __threadfence_block();
        __syncthreads(); // the backtrace reads what every thread stored to global memory; the ring is dead
        if (tid < 64) {
            WfaOut o;
            o.blast_score = 0;
            o.r.status = res.status;
            o.r.score = res.status == 3 ? res.score : 0;
            o.r.nops = 0;
            o.r.qbegin = o.r.qend = o.r.tbegin = o.r.tend = 0;
            o.r.align_len = o.r.matches = o.r.gaps = o.r.gap_regions = 0;
            if (res.status == 0) {
                BtLds &btl = *(BtLds *)ring_raw;
                const int nops = bt_walk(hdr2, bt, res.score, w.tlen - w.qlen, bt + arena_stride - 16, arena_stride - 16 - ((res.used + 15) & ~15), &btl, lane);
                __threadfence_block(); // lane 0's operation bytes are visible to the other lanes of this wavefront
                LDS_WAVE_SYNC();
                if (nops < 0) {
                    o.r.status = 1;
                } else {
                    WfaWin Q, T;
                    Q.buf = L.qbuf;
                    Q.src = w.q;
                    Q.len = w.qlen;
                    Q.w0 = WIN ? -(1 << 24) : 0; // (WIN: nothing counts as resident, the replay's first step loads its windows)
                    T.buf = L.tbuf;
                    T.src = w.t;
                    T.len = w.tlen;
                    T.w0 = WIN ? -(1 << 24) : 0;
                    bt_replay<WIN>(bt + arena_stride - 16 - nops, nops, Q, T, w.qlen, w.tlen, want_ops ? ops_pool + w.ops_off : nullptr, w.ops_cap, lane,
                                     res.score, &o.r, &o.blast_score);
                }
            }
            if (lane == 0) {
                out[i] = o;
                sh_x = atomicAdd(queue, 1u);
            }
        }
    }
}

typedef void (*WfaMwFn)(const WfaIn *, int64_t, const int32_t *, int64_t, int32_t *, int64_t, uint8_t *, int64_t, uint64_t *, unsigned int *, int,
                        int, WfaOut *);
static WfaMwFn wfa_mw_fn(int ncw, bool win) {
    if (win) return ncw == 4 ? k_wfa_mw<4, true> : ncw == 1 ? k_wfa_mw<1, true> : k_wfa_mw<2, true>;
    return ncw == 4 ? k_wfa_mw<4, false> : ncw == 1 ? k_wfa_mw<1, false> : k_wfa_mw<2, false>;
}

} // namespace lm

#define CK(x)                                                                                          \
    do {                                                                                               \
        hipError_t e_ = (x);                                                                           \
        if (e_ != hipSuccess) {                                                                        \
            fprintf(stderr, "wfa_mw: %s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            return -1;                                                                                 \
        }                                                                                              \
    } while (0)

struct MwCompare {
    double ms_lean, ms_mw;                                 // average kernel time per launch
    int64_t n, lean_ok, mw_ok, lean_status3, mw_status3, mw_status1, both_ok, mismatches;
    int32_t blocks_lean, blocks_mw, max_width_reported, pad;
};

// seqs: all sequences back to back; problem i aligns [qoff, qoff+qlen) with [toff, toff+tlen).  ncw 2: 512 diagonals against
// k_wfa_lean<8>, ncw 4: 1024 diagonals against k_wfa_lean<16>
extern "C" int mw_compare(const uint8_t *seqs, int64_t nbytes, const int64_t *qoff, const int32_t *qlen, const int64_t *toff, const int32_t *tlen,
                          int64_t n, int ncw, int win, int reps, MwCompare *res) {
    using namespace lm;
    memset(res, 0, sizeof *res);
    res->n = n;
    uint8_t *d_seq = nullptr;
    CK(hipMalloc(&d_seq, (size_t)nbytes + 64));
    CK(hipMemset(d_seq, 'A', (size_t)nbytes + 64));
    CK(hipMemcpy(d_seq, seqs, (size_t)nbytes, hipMemcpyHostToDevice));
    std::vector<WfaIn> in((size_t)n);
    int64_t ops_tot = 0, lmax = 1;
    int wmax = 1;
    std::vector<std::pair<int64_t, int32_t>> ord;
    for (int64_t i = 0; i < n; i++) {
        WfaIn &w = in[i];
        memset(&w, 0, sizeof w);
        w.q = d_seq + qoff[i];
        w.t = d_seq + toff[i];
        w.qlen = qlen[i];
        w.tlen = tlen[i];
        const int64_t L = (int64_t)qlen[i] + tlen[i];
        w.ops_off = ops_tot;
        w.ops_cap = (int32_t)(L + 2);
        ops_tot += L + 2;
        lmax = std::max(lmax, L);
        wmax = std::max(wmax, (std::max(qlen[i], tlen[i]) + 15) / 16);
        ord.push_back({-L, (int32_t)i});
    }
    std::sort(ord.begin(), ord.end());
    std::vector<int32_t> todo((size_t)n);
    for (int64_t i = 0; i < n; i++) todo[i] = ord[i].second;
    WfaIn *d_in = nullptr;
    int32_t *d_todo = nullptr;
    WfaOut *d_out[2] = {nullptr, nullptr};
    uint64_t *d_ops[2] = {nullptr, nullptr};
    unsigned int *d_queue = nullptr;
    CK(hipMalloc(&d_in, sizeof(WfaIn) * n));
    CK(hipMalloc(&d_todo, sizeof(int32_t) * n));
    CK(hipMalloc(&d_queue, 64));
    for (int k = 0; k < 2; k++) {
        CK(hipMalloc(&d_out[k], sizeof(WfaOut) * n));
        CK(hipMemset(d_out[k], 0xff, sizeof(WfaOut) * n));
        CK(hipMalloc(&d_ops[k], sizeof(uint64_t) * (ops_tot + 16)));
    }
    CK(hipMemcpy(d_in, in.data(), sizeof(WfaIn) * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_todo, todo.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice));
    const int64_t s_expect = (int64_t)(5.0 * 0.13 * (double)lmax) + 2048;
    const int64_t smax = std::min<int64_t>(8 * lmax + 64, s_expect);
    const int64_t entries = smax / 2 + 4;
    const int W = 256 * ncw, nc = 4 * ncw;
    int64_t bytes = (smax / 2 + 2) * W + 2 * lmax + 4096;
    bytes = std::max<int64_t>(bytes, 65536) & ~(int64_t)15;
    int device = 0, cus = 256;
    CK(hipGetDevice(&device));
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int which = 0; which < 2; which++) { // 0: the product's single-wavefront kernel, 1: four wavefronts per alignment
        const size_t lds = win ? 0 : (size_t)(2 * (wmax + 2)) * sizeof(uint32_t);
        int nb = 0, nblocks = 0;
        if (which == 0) {
            nblocks = (int)std::min<int64_t>(n, std::max(256, wfa_resident_blocks(device, wmax, nc, win != 0)));
        } else {
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)wfa_mw_fn(ncw, win != 0), MW_THREADS, lds) != hipSuccess || nb < 1) nb = 1;
            nblocks = (int)std::min<int64_t>(n, (int64_t)nb * cus);
        }
        int32_t *hdr = nullptr;
        uint8_t *arena = nullptr;
        CK(hipMalloc(&hdr, sizeof(int32_t) * (size_t)(entries * 2) * nblocks + 64));
        CK(hipMalloc(&arena, (size_t)bytes * nblocks + 64));
        float tot = 0;
        for (int rep = 0; rep < reps + 1; rep++) {
            CK(hipMemsetAsync(d_queue, 0, sizeof(unsigned int), st));
            CK(hipEventRecord(e0, st));
            if (which == 0)
                launch_wfa(st, d_in, n, d_todo, n, nblocks, hdr, entries * 2, arena, bytes, d_ops[0], d_queue, wmax, 1, d_out[0], nc, win != 0);
            else
                hipLaunchKernelGGL(wfa_mw_fn(ncw, win != 0), dim3(nblocks), dim3(MW_THREADS), lds, st, d_in, n, d_todo, n, hdr, entries * 2, arena, bytes, d_ops[1],
                                   d_queue, wmax, 1, d_out[1]);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0) tot += ms;
        }
        (which == 0 ? res->ms_lean : res->ms_mw) = tot / reps;
        (which == 0 ? res->blocks_lean : res->blocks_mw) = nblocks;
        CK(hipFree(hdr));
        CK(hipFree(arena));
    }
    std::vector<WfaOut> o0((size_t)n), o1((size_t)n);
    std::vector<uint64_t> p0((size_t)ops_tot), p1((size_t)ops_tot);
    CK(hipMemcpy(o0.data(), d_out[0], sizeof(WfaOut) * n, hipMemcpyDeviceToHost));
    CK(hipMemcpy(o1.data(), d_out[1], sizeof(WfaOut) * n, hipMemcpyDeviceToHost));
    CK(hipMemcpy(p0.data(), d_ops[0], sizeof(uint64_t) * ops_tot, hipMemcpyDeviceToHost));
    CK(hipMemcpy(p1.data(), d_ops[1], sizeof(uint64_t) * ops_tot, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n; i++) {
        const bool a = o0[i].r.status == 0 || o0[i].r.status == 2, b = o1[i].r.status == 0 || o1[i].r.status == 2;
        res->lean_ok += a;
        res->mw_ok += b;
        res->mw_status3 += o1[i].r.status == 3;
        res->mw_status1 += o1[i].r.status == 1;
        res->lean_status3 += o0[i].r.status == 3;
        if (o1[i].r.status == 3 && o1[i].r.score > res->max_width_reported) res->max_width_reported = o1[i].r.score;
        if (a != b || o0[i].r.status != o1[i].r.status) {
            if (res->mismatches < 5)
                fprintf(stderr, "wfa_mw: problem %lld (%d x %d): status %d vs %d (score field %d vs %d)\n", (long long)i, in[i].qlen, in[i].tlen,
                        o0[i].r.status, o1[i].r.status, o0[i].r.score, o1[i].r.score);
            res->mismatches++;
            continue;
        }
        if (a && b) {
            res->both_ok++;
            bool same = memcmp(&o0[i], &o1[i], sizeof(WfaOut)) == 0;
            for (int j = 0; same && j < o0[i].r.nops; j++) same = p0[in[i].ops_off + j] == p1[in[i].ops_off + j];
            if (!same) {
                if (res->mismatches < 5)
                    fprintf(stderr, "wfa_mw: problem %lld (%d x %d) differs: score %d vs %d, nops %d vs %d\n", (long long)i, in[i].qlen, in[i].tlen,
                            o0[i].r.score, o1[i].r.score, o0[i].r.nops, o1[i].r.nops);
                res->mismatches++;
            }
        }
    }
    hipFree(d_seq);
    hipFree(d_in);
    hipFree(d_todo);
    hipFree(d_queue);
    for (int k = 0; k < 2; k++) {
        hipFree(d_out[k]);
        hipFree(d_ops[k]);
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipStreamDestroy(st);
    return 0;
}
