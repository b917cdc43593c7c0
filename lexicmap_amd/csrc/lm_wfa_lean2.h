// lm_wfa_lean2.h - device side of wfa_lean2_fwd.h : k_wfa_lean2<NC, RT, WIN>,
// the single-wavefront WFA kernel (persistent wavefronts over a queue; the sequences 2-bit packed in LDS, whole or through sliding windows;
// bt_walk / bt_replay of lm_kernels.hip) with the forward pass of lm_wfa_lean2_fwd.h.  8 * seq_words + 20 bytes of dynamic LDS for the
// whole-sequence form.  Included by lm_wfa_dev.h (whose WR_* macros it uses) inside namespace lm.  It replaced k_wfa_lean in
// round 5 (C3 12.1 -> 9.85 s per step on one resident index); the forward pass is also checked on the host SIMT emulator.
#pragma once

#define WR_WAVE_SYNC() LDS_WAVE_SYNC()
#define WR_LDS __attribute__((address_space(3)))
// the wave mask of a predicate as the compare wrote it (__ballot goes through v_cndmask + v_cmp_ne whenever the predicate is
// not a single compare)
#undef WR_BALLOT
#define WR_BALLOT(p) __builtin_amdgcn_ballot_w64(p)
// minimum over the wavefront, result uniform; all 64 lanes active.  The DPP source selection on the minimum itself (six
// v_min_i32_dpp) instead of six v_mov_b32_dpp + six v_min_i32; two wait states between a write and the DPP read of it.
__device__ __forceinline__ int l2_wave_min_i32(int v) {
#ifdef L2_NO_ASM /* -DL2_NO_ASM: the builtin forms of everything written in assembly here (first thing to try if the GPU disagrees) */
    return wave_min_i32(v);
#else
    asm("s_nop 1\n\tv_min_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
        : "+v"(v));
    return __builtin_amdgcn_readlane(v, 63);
#endif
}
#undef WR_WAVE_MIN_I32
#define WR_WAVE_MIN_I32(v) l2_wave_min_i32(v)
// first set bit / leading zeros of a wave mask on the scalar unit, -1 for an empty mask (the C builtins leave that case undefined)
__device__ __forceinline__ int l2_sff1(unsigned long long m) {
#ifdef L2_NO_ASM
    return m ? __builtin_ctzll(m) : -1;
#else
    int r;
    asm("s_ff1_i32_b64 %0, %1" : "=s"(r) : "s"(m));
    return r;
#endif
}
__device__ __forceinline__ int l2_sflb(unsigned long long m) {
#ifdef L2_NO_ASM
    return m ? __builtin_clzll(m) : -1;
#else
    int r;
    asm("s_flbit_i32_b64 %0, %1" : "=s"(r) : "s"(m));
    return r;
#endif
}
#define WR_FF1(x) l2_sff1(x)
#define WR_FLB(x) l2_sflb(x)
#define WR_READLANE(v, l) __builtin_amdgcn_readlane((v), (l)) /* `l` is wave-uniform */
#define WR_ALIGNBIT(hi, lo, sh) __builtin_amdgcn_alignbit((hi), (lo), (sh))
// -2 pos modulo 32 (positions are below 2^24): one full-rate v_mul_u32_u24 the compiler cannot turn back into v_mul_lo_u32
__device__ __forceinline__ uint32_t l2_neg2(int pos) {
#ifdef L2_NO_ASM
    return 30u * (uint32_t)pos;
#else
    uint32_t r;
    asm("v_mul_u32_u24 %0, %1, 30" : "=v"(r) : "v"(pos));
    return r;
#endif
}
#define WR_NEG2(pos) l2_neg2(pos)

#include "lm_wfa_lean2_fwd.h"

// WPE: wavefronts per SIMD the register allocation is held to (8 = at most 64 VGPRs: the instantiation of the two shortest length
// classes, whose residency the registers bound - 69 VGPRs were 7 wavefronts per SIMD; 1 = as the compiler likes)
template <int NC, typename RT, bool WIN, int MARGIN = L2_SHRINK_MARGIN, int WPE = 1>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE))) void k_wfa_lean2(const WfaIn *__restrict__ in, int64_t n, const int32_t *__restrict__ todo, int64_t ntodo,
                                                   int32_t *__restrict__ hdr_pool, int64_t hdr_stride, uint8_t *__restrict__ arena_pool,
                                                   int64_t arena_stride, uint64_t *__restrict__ ops_pool, unsigned int *__restrict__ queue,
                                                   int seq_words, int want_ops, WfaOut *__restrict__ out, unsigned long long *__restrict__ dbg) {
    constexpr int RING_CELLS = l2_ring_cells<NC>() * (int)sizeof(RT);
    constexpr int BW = RING_CELLS >= (int)sizeof(BtLds) ? BT_WIN : ((RING_CELLS - 512 - 32) & ~15);
    static_assert(BW >= 2 * 64 * NC, "the walk's window holds at least two rows");
    constexpr int RING_BYTES = RING_CELLS > (int)sizeof(BtLdsT<BW>) ? RING_CELLS : (int)sizeof(BtLdsT<BW>);
    __shared__ __attribute__((aligned(16))) uint8_t ring_raw[RING_BYTES]; // the backtrace walk reuses the ring (dead by then)
    BtLdsT<BW> &btl = *(BtLdsT<BW> *)ring_raw;
    __shared__ unsigned int sh_x;
    // WIN: the two sequence windows (one readable word in front of each); otherwise one word, then both whole packed sequences,
    // seq_words + 2 words each, in dynamic LDS (launcher: 8 * seq_words + 20 bytes)
    __shared__ uint32_t qwin_buf[WIN ? L2_WINW + 3 : 1], twin_buf[WIN ? L2_WINW + 3 : 1];
    extern __shared__ uint32_t seq_lds[];
    const int lane = threadIdx.x;
    int32_t *hdr2 = hdr_pool + (int64_t)blockIdx.x * hdr_stride;
    uint8_t *bt = arena_pool + (int64_t)blockIdx.x * arena_stride;
    const int max_score = (int)(hdr_stride / 2 - 2) * 2;
    if (NC >= 16)
        __builtin_amdgcn_s_setprio(3);
    else if (NC >= 8)
        __builtin_amdgcn_s_setprio(2);
    else if (NC >= 4)
        __builtin_amdgcn_s_setprio(1);
    (void)dbg;
    if (lane == 0) sh_x = atomicAdd(queue, 1u);
    while (true) {
        LDS_WAVE_SYNC();
        const unsigned int x = (unsigned int)__builtin_amdgcn_readfirstlane((int)sh_x);
        LDS_WAVE_SYNC();
        if ((int64_t)x >= ntodo) break;
        const int64_t i = todo ? todo[x] : (int64_t)x;
        if (i < 0 || i >= n) break; // malformed work list
        const WfaIn w = in[i];
        const int plen = w.qlen, tlen = w.tlen;
        WfaWin Q, T; // (bt_replay's view of the packed sequences)
        Q.buf = WIN ? qwin_buf + 1 : seq_lds + 1; // (l2_get16 / l2_win_get32 read one word in front)
        Q.src = w.q;
        Q.len = plen;
        Q.w0 = 0;
        T.buf = WIN ? twin_buf + 1 : seq_lds + 1 + seq_words + 2;
        T.src = w.t;
        T.len = tlen;
        T.w0 = 0;
        L2Res r;
        r.status = 0;
        r.score = 0;
        r.used = 0;
        LDS_WAVE_SYNC(); // the previous alignment is done with the sequences and the ring
        if (!WIN) {
            bool bad = false;
            const int qw = (plen + 15) >> 4, tw = (tlen + 15) >> 4;
            if (qw > seq_words || tw > seq_words) {
                r.status = 3;
            } else {
                for (int j = lane; j < qw; j += 64) Q.buf[j] = pack16(w.q + 16 * j, plen - 16 * j, &bad);
                for (int j = lane; j < tw; j += 64) T.buf[j] = pack16(w.t + 16 * j, tlen - 16 * j, &bad);
                if (lane < 2) {
                    Q.buf[qw + lane] = 0;
                    T.buf[tw + lane] = 0;
                }
                if (__ballot(bad) != 0ull) r.status = 3; // not plain ACGT: the byte-comparing kernel takes it
            }
        }
        if (r.status == 0) {
            L2Prob p;
            p.q = w.q;
            p.t = w.t;
            p.plen = plen;
            p.tlen = tlen;
            p.hdr2 = hdr2;
            p.bt = bt;
            p.arena_cap = (int32_t)(arena_stride - 16); // the window copies of the walk read whole 16-byte chunks
            p.max_score = max_score;
            LDS_WAVE_SYNC();
            wfa_lean2_forward<NC, RT, WIN, MARGIN>(p, (RT *)ring_raw, Q.buf, T.buf, &r);
            Q.w0 = r.qw0; // (WIN: bt_replay moves the windows on from where the pass left them)
            T.w0 = r.tw0;
        }
        __syncthreads(); // the backtrace reads what every lane stored to global memory
        WfaOut o;
        o.blast_score = 0;
        if (r.status != 0) {
            o.r.status = r.status;
            o.r.score = r.status == 3 ? r.score : 0;
            o.r.nops = 0;
            o.r.qbegin = o.r.qend = o.r.tbegin = o.r.tend = 0;
            o.r.align_len = o.r.matches = o.r.gaps = o.r.gap_regions = 0;
        } else {
            const int nops = bt_walk(hdr2, bt, r.score, tlen - plen, bt + arena_stride - 16, arena_stride - 16 - ((r.used + 15) & ~15), &btl, lane);
            __threadfence_block();
            __syncthreads(); // lane 0's operation bytes are visible to the other lanes
            if (nops < 0) {
                o.r.status = 1;
                o.r.score = 0;
                o.r.nops = 0;
                o.r.qbegin = o.r.qend = o.r.tbegin = o.r.tend = 0;
                o.r.align_len = o.r.matches = o.r.gaps = o.r.gap_regions = 0;
            } else {
                bt_replay<WIN>(bt + arena_stride - 16 - nops, nops, Q, T, plen, tlen, want_ops ? ops_pool + w.ops_off : nullptr, w.ops_cap, lane,
                                 r.score, &o.r, &o.blast_score);
            }
        }
        if (lane == 0) {
            out[i] = o;
            sh_x = atomicAdd(queue, 1u);
        }
    }
}
