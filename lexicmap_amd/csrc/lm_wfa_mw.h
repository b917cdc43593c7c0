// lm_wfa_mw.h - the macros under which the WFA forward passes compile on the device (lm_wfa_lean2_fwd.h, lm_wfa_mw2_fwd.h: one
// source for the device and the host SIMT emulator of tests/emu) and the launcher of k_wfa_mw2<NCW, WIN>: the LDS wavefront
// kernel with a WORKGROUP of four wavefronts per alignment (512 / 1024 diagonals; WIN: sequences through sliding windows), for
// the handful of 20-50-kb alignments per round whose wavefronts outgrow 256 diagonals and that every round waits for.
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

#include "lm_wfa_lean2.h"
#include "lm_wfa_mw2.h"

typedef void (*WfaMwFn)(const WfaIn *, int64_t, const int32_t *, int64_t, int32_t *, int64_t, uint8_t *, int64_t, uint64_t *, unsigned int *, int,
                        int, WfaOut *);
static WfaMwFn wfa_mw_fn(int ncw, bool win) {
    if (win) return ncw == 4 ? k_wfa_mw2<4, true> : k_wfa_mw2<2, true>;
    return ncw == 4 ? k_wfa_mw2<4, false> : k_wfa_mw2<2, false>;
}
static size_t wfa_mw_dyn_lds(int seq_words, bool win) { return win ? 0 : (size_t)(2 * (seq_words + 2) + 1) * sizeof(uint32_t); } // (one word in front)
// workgroups of k_wfa_mw2<nc / 4> the device holds at once (nc = 8: 512 diagonals, 16: 1024)
int wfa_mw_resident_blocks(int device, int seq_words, int nc, bool win) {
    int nb = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)wfa_mw_fn(nc / 4, win), MW2_THREADS, wfa_mw_dyn_lds(seq_words, win)) != hipSuccess || nb < 1)
        nb = 1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 1) cus = 256;
    return nb * cus;
}
void launch_wfa_mw(hipStream_t st, const WfaIn *in, int64_t n, const int32_t *todo, int64_t ntodo, int nblocks, int32_t *hdr_pool,
                   int64_t hdr_stride, uint8_t *arena_pool, int64_t arena_stride, uint64_t *ops_pool, unsigned int *queue, int seq_words,
                   int want_ops, WfaOut *out, int nc, bool win) {
    hipLaunchKernelGGL(wfa_mw_fn(nc / 4, win), dim3(nblocks), dim3(MW2_THREADS), wfa_mw_dyn_lds(seq_words, win), st, in, n, todo, ntodo, hdr_pool, hdr_stride,
                       arena_pool, arena_stride, ops_pool, queue, seq_words, want_ops, out);
}
