// lm_wfa_mw.h - k_wfa_mw<NCW>: the LDS wavefront kernel with a WORKGROUP of four wavefronts per alignment (512 / 1024
// diagonals; WIN: sequences through sliding windows), for the 512- and 1024-diagonal passes of the long length classes: a handful of 20-50-kb alignments per round
// that k_wfa_lean<8 / 16> runs at single-wavefront latency (8 / 16 cells per lane) while the round waits.  Four wavefronts
// with 2 / 4 cells per lane run the same score step ~2 x faster (three workgroup barriers per score); the backtrace is
// bt_walk / bt_replay by the first wavefront.  Results identical to k_wfa_lean (same rows, same bytes).
// Included by lm_kernels.hip inside namespace lm, after bt_walk / bt_replay / wave_pkmin_u16.
#pragma once

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

#include "lm_wfa_mw_fwd.h"
#include "lm_wfa_lean2.h"
#include "lm_wfa_mw2.h"

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
    // the handful of long alignments every round waits for: issued ahead of the many short ones sharing the SIMDs (as the
    // single-wavefront 512 / 1024-diagonal passes were)
    __builtin_amdgcn_s_setprio(NCW >= 4 ? 3 : 2);
    if (tid == 0) sh_x = atomicAdd(queue, 1u);
    while (true) {
        __syncthreads();
        const unsigned int x = (unsigned int)__builtin_amdgcn_readfirstlane((int)sh_x);
        __syncthreads();
        if ((int64_t)x >= ntodo) break;
        const int64_t i = todo ? todo[x] : (int64_t)x;
        if (i < 0 || i >= n) break; // malformed work list
        const WfaIn w = in[i];
        MwProb p;
        p.q = w.q;
        p.t = w.t;
        p.plen = w.qlen;
        p.tlen = w.tlen;
        p.hdr2 = hdr2;
        p.bt = bt;
        p.arena_cap = (int32_t)(arena_stride - 16 < 2000000000 ? arena_stride - 16 : 2000000000);
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
static WfaMwFn wfa_mw_fn(int ncw, bool win, bool lean2 = false) {
    if (lean2) { // the restructured forward pass (lm_wfa_mw2.h)
        if (win) return ncw == 4 ? k_wfa_mw2<4, true> : k_wfa_mw2<2, true>;
        return ncw == 4 ? k_wfa_mw2<4, false> : k_wfa_mw2<2, false>;
    }
    if (win) return ncw == 4 ? k_wfa_mw<4, true> : k_wfa_mw<2, true>;
    return ncw == 4 ? k_wfa_mw<4, false> : k_wfa_mw<2, false>;
}
static size_t wfa_mw_dyn_lds(int seq_words, bool win) { return win ? 0 : (size_t)(2 * (seq_words + 2) + 1) * sizeof(uint32_t); } // (k_wfa_mw2: one word in front)
// workgroups of k_wfa_mw<nc / 4> the device holds at once (nc = 8: 512 diagonals, 16: 1024)
int wfa_mw_resident_blocks(int device, int seq_words, int nc, bool win, bool lean2) {
    int nb = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)wfa_mw_fn(nc / 4, win, lean2), MW_THREADS, wfa_mw_dyn_lds(seq_words, win)) != hipSuccess || nb < 1)
        nb = 1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 1) cus = 256;
    return nb * cus;
}
void launch_wfa_mw(hipStream_t st, const WfaIn *in, int64_t n, const int32_t *todo, int64_t ntodo, int nblocks, int32_t *hdr_pool,
                   int64_t hdr_stride, uint8_t *arena_pool, int64_t arena_stride, uint64_t *ops_pool, unsigned int *queue, int seq_words,
                   int want_ops, WfaOut *out, int nc, bool win, bool lean2) {
    hipLaunchKernelGGL(wfa_mw_fn(nc / 4, win, lean2), dim3(nblocks), dim3(MW_THREADS), wfa_mw_dyn_lds(seq_words, win), st, in, n, todo, ntodo, hdr_pool, hdr_stride,
                       arena_pool, arena_stride, ops_pool, queue, seq_words, want_ops, out);
}
