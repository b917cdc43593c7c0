// lm_pa_chain_bt.h - device side of pa_chain_bt.h : the macros under which the
// wavefront backtrack compiles inside lm_kernels.hip (namespace lm, after lm_pa_chain_dp.h: it uses pcd_wave_max_u64).
// k_pa_chain_wave's backtrack and ClearSubstrPairs marks since round 5 (the lane-0 / global-memory forms lost the A/B and went).
#pragma once

#define PCB_DEV __device__ __forceinline__
#define PCB_LANE ((int)(threadIdx.x & 63))
#define PCB_UNIFORM(x) __builtin_amdgcn_readfirstlane((int)(x))
#define PCB_LDS_SYNC() LDS_WAVE_SYNC()
#define PCB_WAVE_MAX_U64(v) pcd_wave_max_u64(v)

#include "lm_pa_chain_bt_core.h"

#define PCC_DEV __device__ __forceinline__
#define PCC_LANE ((int)(threadIdx.x & 63))
#define PCC_LDS_SYNC() LDS_WAVE_SYNC()
#include "lm_pa_clear_tile.h"
static_assert(sizeof(PccLds) <= sizeof(PcdLds), "the clear tile aliases the DP's ring");
