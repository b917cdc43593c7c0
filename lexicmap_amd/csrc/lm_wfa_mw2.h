// lm_wfa_mw2.h - device side of wfa_mw2_fwd.h : k_wfa_mw2<NCW, WIN>, the workgroup WFA kernel
// (a workgroup of four wavefronts per long alignment; persistent over a queue; bt_walk / bt_replay by the first wavefront)
// with the forward pass of lm_wfa_mw2_fwd.h; dynamic LDS of the
// whole-sequence form 8 * seq_words + 20 bytes.  Included inside namespace lm after lm_wfa_lean2.h.  It replaced k_wfa_mw in round 5;
// forced through every instantiation on the GPU by tests/test_gpu_wfa_lean2.py / test_gpu_wfa_mw.py, and on the host SIMT emulator.
#pragma once

// the minimum over the four lanes of a quad, in each of them (two DPP quad permutations)
__device__ __forceinline__ int l2_quad_min_i32(int v) {
    int x = __builtin_amdgcn_mov_dpp(v, 0xb1, 0xf, 0xf, false); // quad_perm:[1,0,3,2]
    v = x < v ? x : v;
    x = __builtin_amdgcn_mov_dpp(v, 0x4e, 0xf, 0xf, false); // quad_perm:[2,3,0,1]
    return x < v ? x : v;
}
#define WR_QUAD_MIN_I32(v) l2_quad_min_i32(v)

#include "lm_wfa_mw2_fwd.h"

template <int NCW, bool WIN>
__global__ __launch_bounds__(MW2_THREADS) void k_wfa_mw2(const WfaIn *__restrict__ in, int64_t n, const int32_t *__restrict__ todo, int64_t ntodo,
                                                          int32_t *__restrict__ hdr_pool, int64_t hdr_stride, uint8_t *__restrict__ arena_pool,
                                                          int64_t arena_stride, uint64_t *__restrict__ ops_pool, unsigned int *__restrict__ queue,
                                                          int seq_words, int want_ops, WfaOut *__restrict__ out) {
    constexpr int RING_CELLS = mw2_ring_cells<NCW>() * 4;
    constexpr int RING_BYTES = RING_CELLS > (int)sizeof(BtLds) ? RING_CELLS : (int)sizeof(BtLds);
    __shared__ __attribute__((aligned(16))) uint8_t ring_raw[RING_BYTES]; // the backtrace walk reuses the ring (dead by then)
    __shared__ int32_t red[MW2_RED_WORDS];
    __shared__ unsigned int sh_x;
    __shared__ int sh_bad;
    __shared__ uint32_t qwin_buf[WIN ? L2_WINW + 3 : 1], twin_buf[WIN ? L2_WINW + 3 : 1];
    extern __shared__ uint32_t seq_lds[]; // one word, then both whole packed sequences, seq_words + 2 words each
    const int tid = threadIdx.x, lane = tid & 63;
    int32_t *hdr2 = hdr_pool + (int64_t)blockIdx.x * hdr_stride;
    uint8_t *bt = arena_pool + (int64_t)blockIdx.x * arena_stride;
    const int max_score = (int)(hdr_stride / 2 - 2) * 2;
    __builtin_amdgcn_s_setprio(NCW >= 4 ? 3 : 2);
    if (tid == 0) sh_x = atomicAdd(queue, 1u);
    while (true) {
        __syncthreads();
        const unsigned int x = (unsigned int)__builtin_amdgcn_readfirstlane((int)sh_x);
        if (tid == 0) sh_bad = 0;
        __syncthreads();
        if ((int64_t)x >= ntodo) break;
        const int64_t i = todo ? todo[x] : (int64_t)x;
        if (i < 0 || i >= n) break; // malformed work list
        const WfaIn w = in[i];
        uint32_t *const qbuf = WIN ? qwin_buf + 1 : seq_lds + 1, *const tbuf = WIN ? twin_buf + 1 : seq_lds + 1 + seq_words + 2;
        L2Res res;
        res.status = 0;
        res.score = 0;
        res.used = 0;
        res.qw0 = res.tw0 = 0;
        if (!WIN) {
            bool bad = false;
            const int qw = (w.qlen + 15) >> 4, tw = (w.tlen + 15) >> 4;
            if (qw > seq_words || tw > seq_words) {
                res.status = 3;
            } else {
                for (int j = tid; j < qw; j += MW2_THREADS) qbuf[j] = pack16(w.q + 16 * j, w.qlen - 16 * j, &bad);
                for (int j = tid; j < tw; j += MW2_THREADS) tbuf[j] = pack16(w.t + 16 * j, w.tlen - 16 * j, &bad);
                if (tid < 2) {
                    qbuf[qw + tid] = 0;
                    tbuf[tw + tid] = 0;
                }
                if (__ballot(bad) != 0ull && lane == 0) sh_bad = 1;
            }
            __syncthreads();
            if (res.status == 0 && __builtin_amdgcn_readfirstlane(sh_bad) != 0) res.status = 3; // not plain ACGT
        }
        if (res.status == 0) {
            L2Prob p;
            p.q = w.q;
            p.t = w.t;
            p.plen = w.qlen;
            p.tlen = w.tlen;
            p.hdr2 = hdr2;
            p.bt = bt;
            p.arena_cap = (int32_t)(arena_stride - 16 < 2000000000 ? arena_stride - 16 : 2000000000);
            p.max_score = max_score;
            wfa_mw2_forward<NCW, WIN>(p, (int32_t *)ring_raw, qbuf, tbuf, red, &res);
        }
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
                    Q.buf = qbuf;
                    Q.src = w.q;
                    Q.len = w.qlen;
                    Q.w0 = WIN ? -(1 << 24) : 0; // (WIN: nothing counts as resident, the replay's first step loads its windows)
                    T.buf = tbuf;
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
