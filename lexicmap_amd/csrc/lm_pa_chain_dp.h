// lm_pa_chain_dp.h - device side of pa_chain_dp.h for lm_kernels.hip (included inside namespace lm, before k_pa_chain_wave):
// the banded DP of Chainer2 with the recent anchors in an LDS ring and a DPP reduction of the 64-bit (score, ~j) key.
#pragma once

// maximum over the wavefront, left in every lane; all 64 lanes active.  Same DPP steps as wave_min_i32, on both halves.
__device__ __forceinline__ unsigned long long pcd_wave_max_u64(unsigned long long v) {
#define PCD_STEP(DPP)                                                          \
    {                                                                          \
        const int lo = DPP((int)(uint32_t)v), hi = DPP((int)(uint32_t)(v >> 32)); \
        const unsigned long long x = ((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo; \
        v = x > v ? x : v;                                                     \
    }
#define PCD_D1(w) __builtin_amdgcn_mov_dpp((w), 0xb1, 0xf, 0xf, false)
#define PCD_D2(w) __builtin_amdgcn_mov_dpp((w), 0x4e, 0xf, 0xf, false)
#define PCD_D3(w) __builtin_amdgcn_mov_dpp((w), 0x124, 0xf, 0xf, false)
#define PCD_D4(w) __builtin_amdgcn_mov_dpp((w), 0x128, 0xf, 0xf, false)
#define PCD_D5(w) __builtin_amdgcn_update_dpp((w), (w), 0x142, 0xa, 0xf, false)
#define PCD_D6(w) __builtin_amdgcn_update_dpp((w), (w), 0x143, 0xc, 0xf, false)
    PCD_STEP(PCD_D1)
    PCD_STEP(PCD_D2)
    PCD_STEP(PCD_D3)
    PCD_STEP(PCD_D4)
    PCD_STEP(PCD_D5)
    PCD_STEP(PCD_D6)
#undef PCD_STEP
    const uint32_t rl = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, 63), rh = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), 63);
    return ((unsigned long long)rh << 32) | rl;
}

#define PCD_DEV __device__ __forceinline__
#define PCD_LANE ((int)threadIdx.x)
#define PCD_BALLOT(p) __ballot(p)
#define PCD_LDS_SYNC() LDS_WAVE_SYNC()
#define PCD_GLOBAL_FENCE() __threadfence_block()
#define PCD_POPCLL(x) __popcll(x)
#define PCD_FFSLL(x) __ffsll((long long)(x))
#define PCD_WAVE_MAX_U64(v) pcd_wave_max_u64(v)

// one-lane wavefront shift: lane 0 <- the wave-uniform `newv`, lane l <- lane l - 1 (DPP wave_shr:1; lane 0 has no source and keeps `old`)
__device__ __forceinline__ int pcd_shift_in(int newv, int v) { return __builtin_amdgcn_update_dpp(newv, v, 0x138, 0xf, 0xf, false); }
#define PCD_SHIFT_IN(newv, v) pcd_shift_in((int)(newv), (int)(v))
// signed maximum over the wavefront, in every lane (the DPP steps of wave_min_i32, lm_kernels.hip)
__device__ __forceinline__ int pcd_wave_max_i32(int v) {
    int x = __builtin_amdgcn_mov_dpp(v, 0xb1, 0xf, 0xf, false);
    v = x > v ? x : v;
    x = __builtin_amdgcn_mov_dpp(v, 0x4e, 0xf, 0xf, false);
    v = x > v ? x : v;
    x = __builtin_amdgcn_mov_dpp(v, 0x124, 0xf, 0xf, false);
    v = x > v ? x : v;
    x = __builtin_amdgcn_mov_dpp(v, 0x128, 0xf, 0xf, false);
    v = x > v ? x : v;
    x = __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false);
    v = x > v ? x : v;
    x = __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false);
    v = x > v ? x : v;
    return __builtin_amdgcn_readlane(v, 63);
}
#define PCD_WAVE_MAX_I32(v) pcd_wave_max_i32(v)
#define PCD_CLZLL(x) __clzll((long long)(x))

#include "lm_pa_chain_dp_core.h"
