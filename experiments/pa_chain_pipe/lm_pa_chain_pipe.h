// experiments/pa_chain_pipe/lm_pa_chain_pipe.h - device side of pa_chain_pipe.h (STAGED for round 5): k_pa_chain_pipe, a
// workgroup of PCP_NW wavefronts per LONG chaining window (n >= a few hundred anchors; the short ones - nearly all windows -
// stay with k_pa_chain_wave, one wavefront each).  Included inside namespace lm after k_pa_chain_wave (lm_kernels.hip): it
// reuses lm_unpack_anchor / lm_trim / the backtrack of lm_run_chain2's second half through the same scratch pools.
// NOT run on a GPU yet: compiled for gfx950 (experiments/pa_chain_pipe/compile_check.hip), the DP itself checked on the
// host SIMT emulator.  What the first GPU run has to confirm is listed at the end of this file.
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

#include "pa_chain_pipe.h"

// One workgroup (PCP_NW * 64 threads) per window of `long_tasks` (the tasks with more than LM_PA_PIPE_MIN anchors, listed by
// the host or by a compaction kernel).  Same inputs, scratch pools and outputs as k_pa_chain_wave.
__global__ __launch_bounds__(PCP_NW * 64) void k_pa_chain_pipe(const uint64_t *__restrict__ B, const int64_t *__restrict__ pa_off,
                                                                const int32_t *__restrict__ long_tasks, int nlong, int K, LmChain2Opt opt,
                                                                LmSub *__restrict__ subs_pool, uint8_t *__restrict__ marks_pool,
                                                                uint64_t *__restrict__ msi_pool, int32_t *__restrict__ stack_pool,
                                                                LmChain2 *__restrict__ out_pool, int32_t *__restrict__ out_n,
                                                                int32_t *__restrict__ clr_n, int qbits, int tbits) {
    constexpr int T = PCP_NW * 64;
    __shared__ PcpLds pl;
    __shared__ int sh_n, sh_start, sh_w;
    __shared__ int sh_cnt[PCP_NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int li = blockIdx.x; li < nlong; li += gridDim.x) {
        const int64_t ti = long_tasks[li];
        const int64_t o = pa_off[ti];
        int n = (int)(pa_off[ti + 1] - o);
        LmSub *sb = subs_pool + o;
        uint8_t *marks = marks_pool + o;
        uint64_t *msi = msi_pool + o;
        LmChain2 *res = out_pool + o;
        __syncthreads();
        for (int i = tid; i < n; i += T) { // unpack (as k_pa_chain_wave)
            const uint64_t v = B[o + i];
            if (qbits > 0) {
                LmSub u;
                u.qbegin = (int32_t)((v >> (8 + tbits)) & ((1ull << qbits) - 1ull));
                u.len = (uint8_t)(32 - (int)((v >> (2 + tbits)) & 63));
                u.tbegin = (int32_t)((v >> 2) & ((1ull << tbits) - 1ull));
                u.qrc = (uint8_t)((v >> 1) & 1);
                u.trc = (uint8_t)(v & 1);
                u.pad = 0;
                sb[i] = u;
            } else {
                sb[i] = lm_unpack_anchor(v);
            }
        }
        __threadfence_block();
        __syncthreads();
        // ---- ClearSubstrPairs: a thread per anchor (an anchor's mark depends on the original list only) ----
        for (int i = tid; i < n; i += T) {
            uint8_t mk = 0;
            if (i >= 1) {
                const LmSub v = sb[i];
                const int32_t vqend = v.qbegin + v.len;
                int32_t upbound = vqend - K;
                if (upbound < 0) upbound = 0;
                const int32_t vtend = v.tbegin + v.len;
                int lo = 0, hi = i;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (sb[mid].qbegin < upbound)
                        lo = mid + 1;
                    else
                        hi = mid;
                }
                for (int j = lo; j < i; j++) {
                    const LmSub p = sb[j];
                    if (vqend <= p.qbegin + p.len && v.tbegin >= p.tbegin && vtend <= p.tbegin + p.len) {
                        mk = 1;
                        break;
                    }
                }
            }
            marks[i] = mk;
        }
        __threadfence_block();
        __syncthreads();
        // ordered in-place compaction, T anchors per pass (wave counts through LDS)
        if (tid == 0) sh_w = 0;
        __syncthreads();
        for (int c = 0; c < n; c += T) {
            const int i = c + tid;
            const bool keep = i < n && !marks[i];
            LmSub v;
            if (keep) v = sb[i];
            const unsigned long long bal = __ballot(keep);
            if (lane == 0) sh_cnt[wave] = __popcll(bal);
            __syncthreads(); // (also: every thread has read its anchor before anybody writes)
            int before = sh_w + __popcll(bal & ((1ull << lane) - 1ull));
            int total = 0;
            for (int w2 = 0; w2 < PCP_NW; w2++) {
                if (w2 < wave) before += sh_cnt[w2];
                total += sh_cnt[w2];
            }
            if (keep) sb[before] = v;
            __threadfence_block();
            __syncthreads();
            if (tid == 0) sh_w += total;
            __syncthreads();
        }
        // ---- TrimSubStrPairs (thread 0; it stops after a few anchors) ----
        if (tid == 0) {
            int start = 0;
            sh_n = lm_trim(sb, sh_w, 100.0f, &start);
            sh_start = start;
            clr_n[ti] = sh_n;
        }
        __syncthreads();
        n = sh_n;
        const LmSub *a_ = sb + sh_start;
        if (n <= 1) { // (a long window that clears down to nothing: the wavefront kernel's small cases)
            if (tid == 0) out_n[ti] = n <= 0 ? 0 : lm_run_chain2(a_, 1, opt, msi, stack_pool + 2 * o + 4 * ti, res);
            continue;
        }
        long long M = 0;
        int Mi = 0;
        pa_chain_dp_pipe(a_, n, opt, msi, &pl, &M, &Mi);
        __threadfence_block();
        __syncthreads();
        // ---- backtrack with the explicit region stack: identical to k_pa_chain_wave's (thread 0) ----
        if (tid == 0) out_n[ti] = lm_chain2_backtrack(a_, n, opt, msi, M, Mi, stack_pool + 2 * o + 4 * ti, res);
    }
}

// To do at integration (round 5):
//  * lm_chain2_backtrack: k_pa_chain_wave's lane-0 block after its DP, factored out as a function both kernels call;
//  * the list of long tasks (n > ~512 anchors): one pass over pa_off after k_pa_task_off_sorted; k_pa_chain_wave skips them;
//  * first GPU run: test_pseudoalign_parity, tests/test_gpu_c4c5.py, tests/test_gpu_longreads.py; then a C4 shard line
//    (target: k_pa_chain 130 -> < 40 ms per launch) - the spin-waits on `done` are the one thing the emulator cannot time.
