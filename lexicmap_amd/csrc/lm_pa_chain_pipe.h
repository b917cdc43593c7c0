// lm_pa_chain_pipe.h - device side of pa_chain_pipe.h : k_pa_chain_pipe, a
// workgroup of PCP_NW wavefronts for the Chainer2 DP of a LONG chaining window.  k_pa_chain_wave (one wavefront per window)
// keeps unpacking, ClearSubstrPairs and TrimSubStrPairs of every window and the whole of the short ones - nearly all of them;
// a window with more than `pipe_min` anchors left after the trim is handed over (clr_n[ti] = anchors, out_n[ti] = first anchor,
// the task appended to a list) and gets its DP here and the backtrack of lm_run_chain2's second half (lm_chain2_backtrack:
// the block k_pa_chain_wave ran on lane 0, factored out by tools/adopt_pa_chain_pipe.py).  Included inside namespace lm after
// k_pa_chain_wave.  NOT run on a GPU yet: compiled for gfx950, the DP checked on the host SIMT emulator
// (tests/test_pa_chain_pipe_emulated_cpu.py).
#pragma once

#define PCP_DEV __device__ __forceinline__
#define PCP_TID ((int)threadIdx.x)
#define PCP_BALLOT(p) __ballot(p)
#define PCP_WAVE_SYNC() LDS_WAVE_SYNC()
#define PCP_BARRIER() __syncthreads()
#define PCP_POPCLL(x) __popcll(x)
#define PCP_FFSLL(x) __ffsll((long long)(x))
#define PCP_WAVE_MAX_U64(v) pcd_wave_max_u64(v)
#define PCP_BCAST32(v, l) ((uint32_t)__builtin_amdgcn_readlane((int)(v), (l))) /* `l` is wave-uniform */
#define PCP_LOAD_DONE(p) __builtin_amdgcn_readfirstlane(*(volatile int *)(p))
// LDS operations of one wavefront complete in program order: the score written before is visible to whoever sees the counter
#define PCP_STORE_DONE(p, v) (*(volatile int *)(p) = (v))
#define PCP_SPIN() __builtin_amdgcn_s_sleep(1)
#define PCP_GLOBAL_FENCE() __threadfence_block()
#define PCP_LOAD_MSI(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

#include "lm_pa_chain_pipe_dp.h"

__global__ __launch_bounds__(PCP_NW * 64) void k_pa_chain_pipe(const int64_t *__restrict__ pa_off, const int32_t *__restrict__ long_tasks,
                                                                const unsigned int *__restrict__ nlong_p, LmChain2Opt opt,
                                                                const LmSub *__restrict__ subs_pool, uint64_t *__restrict__ msi_pool,
                                                                int32_t *__restrict__ stack_pool, LmChain2 *__restrict__ out_pool,
                                                                int32_t *__restrict__ out_n, const int32_t *__restrict__ clr_n, int bt_wave) {
    __shared__ PcpLds pl;
    __shared__ PcbLds pcb;
    const int tid = threadIdx.x;
    const unsigned int nlong = *nlong_p;
    for (unsigned int li = blockIdx.x; li < nlong; li += gridDim.x) {
        const int64_t ti = long_tasks[li];
        const int64_t o = pa_off[ti];
        const int n = clr_n[ti], start = out_n[ti]; // (left by k_pa_chain_wave)
        const LmSub *a_ = subs_pool + o + start;
        uint64_t *msi = msi_pool + o;
        __syncthreads(); // the previous window is done with the LDS strip (and has read out_n / clr_n)
        long long M = 0;
        int Mi = 0;
        pa_chain_dp_pipe(a_, n, opt, msi, &pl, &M, &Mi);
        __threadfence_block();
        __syncthreads(); // every wavefront's msi[] entries are visible to thread 0
        if (bt_wave) { // the first wavefront (experiments/pa_chain_bt)
            if (tid < 64) {
                const int no = pa_chain_backtrack_wave(a_, n, opt, msi, M, Mi, stack_pool + 2 * o + 4 * ti, out_pool + o, &pcb);
                if (tid == 0) out_n[ti] = no;
            }
        } else if (tid == 0) {
            out_n[ti] = lm_chain2_backtrack(a_, n, opt, msi, M, Mi, stack_pool + 2 * o + 4 * ti, out_pool + o);
        }
    }
}
