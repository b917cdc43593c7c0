// lm_wfa_dev.h - the macros under which the WFA forward pass (lm_wfa_lean2_fwd.h: one source for the device and the host SIMT
// emulator of tests/emu) compiles on the device.  Included by lm_kernels.hip inside namespace lm, after bt_walk / bt_replay.
// (Round 4-5 also had k_wfa_mw2 here: a WORKGROUP of four wavefronts per alignment for the 512 / 1024-diagonal passes.  With the
// flavours of k_wfa_lean2 - round 6 - the single-wavefront kernel does those passes as fast or faster: c3mini 310 vs 310 ms per
// step, C3 8.0 vs 8.0 s, C4 shard 1.18 vs 1.25 s - profiles/r06_c3mini_mw_ab.json, r06_c3_ab_mw_lanes.json,
// r06_c4_shard0_of_4_mw_ab.json.  Removed.)
#pragma once

#define WR_DEV __device__ __forceinline__
#define WR_TID ((int)threadIdx.x)
#define WR_BALLOT(p) __ballot(p)
#define WR_UNIFORM(x) __builtin_amdgcn_readfirstlane((int)(x))
#define WR_CLZ(x) __clz((int)(x))
#define WR_WAVE_MIN_I32(v) wave_min_i32(v)
#define WR_NULL_OFF LM_NULL_OFF

#include "lm_wfa_lean2.h"
