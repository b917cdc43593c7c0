// wfa_row.hip - EXPERIMENT (staged for the next round; not linked into liblexicmap_hip.so): the WFA kernel with four
// alignments per wavefront (wfa_row_fwd.h) beside the product's k_wfa_lean on the same problems: results compared struct by
// struct and operation by operation, both timed with HIP events.  One translation unit with the product's kernels (included
// as text: bt_walk / bt_replay / launch_wfa are reused unchanged); builds into experiments/wfa_row/libwfa_row_exp.so.
#include "../../lexicmap_amd/csrc/lm_kernels.hip"

#include <stdio.h>
#include <string.h>

#include <vector>

namespace lm {

#define WR_DEV __device__ __forceinline__
#define WR_LANE ((int)threadIdx.x)
#define WR_BALLOT(p) __ballot(p)
#define WR_SHFL(v, src) ((uint32_t)__shfl((int)(v), (src), 64))
#define WR_LDS_SYNC() LDS_WAVE_SYNC()
#define WR_CLZ(x) __clz((int)(x))
#define wr_pk_min_u16 pk_min_u16
// minimum over the 16 lanes of a row, left in every lane of the row: pairs, quads, then the quads by rotation inside the row
__device__ __forceinline__ int wr_row_min_i32(int v) {
    int x;
    x = __builtin_amdgcn_mov_dpp(v, 0xb1, 0xf, 0xf, false); // quad_perm:[1,0,3,2]
    v = x < v ? x : v;
    x = __builtin_amdgcn_mov_dpp(v, 0x4e, 0xf, 0xf, false); // quad_perm:[2,3,0,1]
    v = x < v ? x : v;
    x = __builtin_amdgcn_mov_dpp(v, 0x124, 0xf, 0xf, false); // row_ror:4
    v = x < v ? x : v;
    x = __builtin_amdgcn_mov_dpp(v, 0x128, 0xf, 0xf, false); // row_ror:8
    v = x < v ? x : v;
    return v;
}
__device__ __forceinline__ uint32_t wr_row_pkmin_u16(uint32_t v) {
    uint32_t x;
    x = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xb1, 0xf, 0xf, false);
    v = pk_min_u16(x, v);
    x = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4e, 0xf, 0xf, false);
    v = pk_min_u16(x, v);
    x = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x124, 0xf, 0xf, false);
    v = pk_min_u16(x, v);
    x = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xf, 0xf, false);
    v = pk_min_u16(x, v);
    return v;
}
#define WR_ROW_MIN_I32(v) wr_row_min_i32(v)
#define WR_ROW_PKMIN_U16(v) wr_row_pkmin_u16(v)
#define WR_NULL_OFF LM_NULL_OFF

#include "wfa_row_fwd.h"

// Persistent wavefronts; each pops FOUR problems of the queue at a time (neighbours in the cost-ordered queue: similar length
// and divergence), runs their forward passes side by side, then the backtrace of each by the whole wavefront (bt_walk /
// bt_replay of k_wfa_lean, unchanged).  hdr_stride / arena_stride are per ROW: a workgroup owns four of each.
#ifndef WR_WAVES_PER_EU
#define WR_WAVES_PER_EU 4
#endif
template <int NCR>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(NCR <= 4 ? WR_WAVES_PER_EU : 3))) void k_wfa_row4(const WfaIn *__restrict__ in, int64_t n, const int32_t *__restrict__ todo, int64_t ntodo,
                                                  int32_t *__restrict__ hdr_pool, int64_t hdr_stride, uint8_t *__restrict__ arena_pool,
                                                  int64_t arena_stride, uint64_t *__restrict__ ops_pool, unsigned int *__restrict__ queue,
                                                  int seq_words, int want_ops, WfaOut *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint8_t btl_raw[sizeof(BtLds)];
    BtLds &btl = *(BtLds *)btl_raw;
    __shared__ unsigned int sh_x;
    extern __shared__ uint32_t seq_lds[]; // 8 x (seq_words + 2) words: Q and T of the four rows
    const int lane = threadIdx.x, r = lane >> 4;
    const int sw2 = seq_words + 2;
    const int max_score = (int)(hdr_stride / 2 - 2) * 2;
    if (lane == 0) sh_x = atomicAdd(queue, 4u);
    while (true) {
        LDS_WAVE_SYNC();
        const unsigned int x = (unsigned int)__builtin_amdgcn_readfirstlane((int)sh_x);
        LDS_WAVE_SYNC();
        if ((int64_t)x >= ntodo) break;
        const int64_t xi = (int64_t)x + r;
        int idx = -1;
        if (xi < ntodo) idx = todo ? todo[xi] : (int)xi;
        if (idx < 0 || idx >= n) idx = -1;
        WrRow p;
        p.q = p.t = (const uint8_t *)in; // never read for an empty row
        p.plen = p.tlen = 0;
        if (idx >= 0) {
            const WfaIn w = in[idx];
            p.q = w.q;
            p.t = w.t;
            p.plen = w.qlen;
            p.tlen = w.tlen;
        }
        p.hdr2 = hdr_pool + ((int64_t)blockIdx.x * 4 + r) * hdr_stride;
        p.bt = arena_pool + ((int64_t)blockIdx.x * 4 + r) * arena_stride;
        p.arena_cap = (int32_t)(arena_stride - 16);
        p.max_score = max_score;
        p.qbuf = seq_lds + (2 * r) * sw2;
        p.tbuf = seq_lds + (2 * r + 1) * sw2;
        p.valid = idx >= 0;
        WrRes res;
        wfa_row4_forward<NCR>(p, seq_words, &res);
        __syncthreads(); // the backtrace reads what the lanes stored to global memory
#pragma unroll 1
        for (int rr = 0; rr < 4; rr++) {
            const int i_r = __builtin_amdgcn_readfirstlane(__shfl(idx, 16 * rr, 64));
            const int st = __builtin_amdgcn_readfirstlane(__shfl(res.status, 16 * rr, 64));
            const int sc = __builtin_amdgcn_readfirstlane(__shfl(res.score, 16 * rr, 64));
            const int used = __builtin_amdgcn_readfirstlane(__shfl(res.used, 16 * rr, 64));
            if (i_r < 0) continue;
            const WfaIn w = in[i_r];
            WfaOut o;
            o.blast_score = 0;
            o.r.status = st;
            o.r.score = st == 3 ? sc : 0;
            o.r.nops = 0;
            o.r.qbegin = o.r.qend = o.r.tbegin = o.r.tend = 0;
            o.r.align_len = o.r.matches = o.r.gaps = o.r.gap_regions = 0;
            if (st == 0) {
                const int32_t *hdr2 = hdr_pool + ((int64_t)blockIdx.x * 4 + rr) * hdr_stride;
                uint8_t *bt = arena_pool + ((int64_t)blockIdx.x * 4 + rr) * arena_stride;
                const int nops = bt_walk(hdr2, bt, sc, w.tlen - w.qlen, bt + arena_stride - 16, arena_stride - 16 - ((used + 15) & ~15), &btl, lane);
                __threadfence_block();
                __syncthreads(); // lane 0's operation bytes are visible to the other lanes
                if (nops < 0) {
                    o.r.status = 1;
                } else {
                    WfaWin Q, T;
                    Q.buf = seq_lds + (2 * rr) * sw2;
                    Q.src = w.q;
                    Q.len = w.qlen;
                    Q.w0 = 0;
                    T.buf = seq_lds + (2 * rr + 1) * sw2;
                    T.src = w.t;
                    T.len = w.tlen;
                    T.w0 = 0;
                    bt_replay<false>(bt + arena_stride - 16 - nops, nops, Q, T, w.qlen, w.tlen, want_ops ? ops_pool + w.ops_off : nullptr, w.ops_cap,
                                     lane, sc, &o.r, &o.blast_score);
                }
            }
            if (lane == 0) out[i_r] = o;
        }
        if (lane == 0) sh_x = atomicAdd(queue, 4u);
    }
}

typedef void (*WfaRowFn)(const WfaIn *, int64_t, const int32_t *, int64_t, int32_t *, int64_t, uint8_t *, int64_t, uint64_t *, unsigned int *,
                         int, int, WfaOut *);
static WfaRowFn wfa_row_fn(int ncr) { return ncr == 8 ? k_wfa_row4<8> : ncr == 2 ? k_wfa_row4<2> : k_wfa_row4<4>; }

} // namespace lm

#define CK(x)                                                                                     \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) {                                                                   \
            fprintf(stderr, "wfa_row: %s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            return -1;                                                                            \
        }                                                                                         \
    } while (0)

struct WrCompare {
    double ms_lean, ms_row;         // average kernel time per launch
    int64_t n, lean_ok, row_ok;     // problems, status 0/2 by either kernel
    int64_t row_status3, row_status1, lean_status3;
    int64_t both_ok, mismatches;    // problems both aligned / of those, records or operations that differ
    int32_t blocks_lean, blocks_row, order_by_score;
    double ms_lean_left; // k_wfa_lean on what the row kernel left (status 3) and on what was routed past it
    int64_t n_routed;    // problems sent straight to k_wfa_lean (score per base above the routing threshold)
};

// seqs: all sequences back to back; problem i aligns [qoff, qoff+qlen) with [toff, toff+tlen)
extern "C" int wr_compare(const uint8_t *seqs, int64_t nbytes, const int64_t *qoff, const int32_t *qlen, const int64_t *toff, const int32_t *tlen,
                          int64_t n, int ncr, int reps, int by_score, int route_permille, WrCompare *res) {
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
    std::sort(ord.begin(), ord.end()); // longest first, like the product's queue
    std::vector<int32_t> todo((size_t)n);
    std::vector<int32_t> true_score((size_t)n, 0);
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
    int device = 0, cus = 256;
    CK(hipGetDevice(&device));
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // ---- the product's kernel: 128 diagonals per wavefront, whole sequences in LDS
    {
        const int nc = 2;
        const int resident = wfa_resident_blocks(device, wmax, nc, false);
        const int nblocks = (int)std::min<int64_t>(n, std::max(256, resident));
        int64_t bytes = (smax / 2 + 2) * 64 * nc + 2 * lmax + 4096;
        bytes = std::max<int64_t>(bytes, 65536) & ~(int64_t)15;
        int32_t *hdr = nullptr;
        uint8_t *arena = nullptr;
        CK(hipMalloc(&hdr, sizeof(int32_t) * (size_t)(entries * 2) * nblocks + 64));
        CK(hipMalloc(&arena, (size_t)bytes * nblocks + 64));
        if (by_score) { // the product orders its queue by expected cost (divergence estimate x length): here by the true score
            CK(hipMemsetAsync(d_queue, 0, sizeof(unsigned int), st));
            launch_wfa(st, d_in, n, d_todo, n, nblocks, hdr, entries * 2, arena, bytes, d_ops[0], d_queue, wmax, 1, d_out[0], nc, false);
            CK(hipStreamSynchronize(st));
            std::vector<WfaOut> o((size_t)n);
            CK(hipMemcpy(o.data(), d_out[0], sizeof(WfaOut) * n, hipMemcpyDeviceToHost));
            for (int64_t i = 0; i < n; i++) ord[i] = {-(int64_t)(o[i].r.status == 0 ? o[i].r.score : 1 << 30), (int32_t)i};
            for (int64_t i = 0; i < n; i++) true_score[i] = o[i].r.status == 0 ? o[i].r.score : 1 << 30;
            std::sort(ord.begin(), ord.end());
            for (int64_t i = 0; i < n; i++) todo[i] = ord[i].second;
            CK(hipMemcpy(d_todo, todo.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice));
            res->order_by_score = 1;
        }
        float tot = 0;
        for (int rep = 0; rep < reps + 1; rep++) {
            CK(hipMemsetAsync(d_queue, 0, sizeof(unsigned int), st));
            CK(hipEventRecord(e0, st));
            launch_wfa(st, d_in, n, d_todo, n, nblocks, hdr, entries * 2, arena, bytes, d_ops[0], d_queue, wmax, 1, d_out[0], nc, false);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0) tot += ms;
        }
        res->ms_lean = tot / reps;
        res->blocks_lean = nblocks;
        CK(hipFree(hdr));
        CK(hipFree(arena));
    }
    std::vector<int32_t> routed;
    int64_t n_easy = n;
    // ---- four alignments per wavefront
    {
        const size_t lds = (size_t)8 * (wmax + 2) * sizeof(uint32_t);
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)wfa_row_fn(ncr), 64, lds) != hipSuccess || nb < 1) nb = 8;
        const int nblocks = (int)std::min<int64_t>((n + 3) / 4, (int64_t)nb * cus);
        int64_t bytes = (smax / 2 + 2) * 16 * ncr + 2 * lmax + 4096;
        bytes = std::max<int64_t>(bytes, 16384) & ~(int64_t)15;
        int32_t *hdr = nullptr;
        uint8_t *arena = nullptr;
        CK(hipMalloc(&hdr, sizeof(int32_t) * (size_t)(entries * 2) * nblocks * 4 + 64));
        CK(hipMalloc(&arena, (size_t)bytes * nblocks * 4 + 64));
        // routing (the product knows a divergence estimate per HSP from the pseudo-alignment): what is expected to outgrow the
        // narrow rows goes straight to the 128-diagonal kernel; here the true score per base stands in for the estimate
        std::vector<int32_t> easy;
        for (int64_t i = 0; i < n; i++) {
            const int32_t j = todo[i];
            const bool hard = route_permille > 0 && (int64_t)true_score[j] * 1000 > (int64_t)route_permille * (in[j].qlen + in[j].tlen);
            if (hard)
                routed.push_back(j);
            else
                easy.push_back(j);
        }
        res->n_routed = (int64_t)routed.size();
        n_easy = (int64_t)easy.size();
        CK(hipMemcpy(d_todo, easy.data(), sizeof(int32_t) * easy.size(), hipMemcpyHostToDevice));
        float tot = 0;
        for (int rep = 0; rep < reps + 1; rep++) {
            CK(hipMemsetAsync(d_queue, 0, sizeof(unsigned int), st));
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(wfa_row_fn(ncr), dim3(nblocks), dim3(64), lds, st, d_in, n, d_todo, n_easy, hdr, entries * 2, arena, bytes, d_ops[1],
                               d_queue, wmax, 1, d_out[1]);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0) tot += ms;
        }
        res->ms_row = tot / reps;
        res->blocks_row = nblocks;
        CK(hipFree(hdr));
        CK(hipFree(arena));
    }
    std::vector<WfaOut> o0((size_t)n), o1((size_t)n);
    std::vector<uint64_t> p0((size_t)ops_tot), p1((size_t)ops_tot);
    CK(hipMemcpy(o0.data(), d_out[0], sizeof(WfaOut) * n, hipMemcpyDeviceToHost));
    CK(hipMemcpy(o1.data(), d_out[1], sizeof(WfaOut) * n, hipMemcpyDeviceToHost));
    CK(hipMemcpy(p0.data(), d_ops[0], sizeof(uint64_t) * ops_tot, hipMemcpyDeviceToHost));
    CK(hipMemcpy(p1.data(), d_ops[1], sizeof(uint64_t) * ops_tot, hipMemcpyDeviceToHost));
    { // what the row kernel leaves goes to the 128-diagonal kernel in the product: its time on exactly those problems
        std::vector<int32_t> left(routed);
        for (int64_t i = 0; i < n; i++)
            if (o1[todo[i]].r.status == 3) left.push_back(todo[i]);
        if (!left.empty()) {
            const int nc = 2;
            const int64_t m = (int64_t)left.size();
            const int resident = wfa_resident_blocks(device, wmax, nc, false);
            const int nblocks = (int)std::min<int64_t>(m, std::max(256, resident));
            int64_t bytes = (smax / 2 + 2) * 64 * nc + 2 * lmax + 4096;
            bytes = std::max<int64_t>(bytes, 65536) & ~(int64_t)15;
            int32_t *hdr = nullptr;
            uint8_t *arena = nullptr;
            CK(hipMalloc(&hdr, sizeof(int32_t) * (size_t)(entries * 2) * nblocks + 64));
            CK(hipMalloc(&arena, (size_t)bytes * nblocks + 64));
            CK(hipMemcpy(d_todo, left.data(), sizeof(int32_t) * m, hipMemcpyHostToDevice));
            float tot = 0;
            for (int rep = 0; rep < reps + 1; rep++) {
                CK(hipMemsetAsync(d_queue, 0, sizeof(unsigned int), st));
                CK(hipEventRecord(e0, st));
                launch_wfa(st, d_in, n, d_todo, m, nblocks, hdr, entries * 2, arena, bytes, d_ops[1], d_queue, wmax, 1, d_out[1], nc, false);
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0) tot += ms;
            }
            res->ms_lean_left = tot / reps;
            CK(hipFree(hdr));
            CK(hipFree(arena));
        }
    }
    for (int64_t i = 0; i < n; i++) {
        const bool a = o0[i].r.status == 0 || o0[i].r.status == 2, b = o1[i].r.status == 0 || o1[i].r.status == 2;
        res->lean_ok += a;
        res->row_ok += b;
        res->row_status3 += o1[i].r.status == 3;
        res->row_status1 += o1[i].r.status == 1;
        res->lean_status3 += o0[i].r.status == 3;
        if (a && b) {
            res->both_ok++;
            bool same = memcmp(&o0[i], &o1[i], sizeof(WfaOut)) == 0;
            for (int j = 0; same && j < o0[i].r.nops; j++) same = p0[in[i].ops_off + j] == p1[in[i].ops_off + j];
            if (!same) {
                if (res->mismatches < 5)
                    fprintf(stderr, "wfa_row: problem %lld (%d x %d) differs: score %d vs %d, nops %d vs %d, status %d vs %d\n", (long long)i,
                            in[i].qlen, in[i].tlen, o0[i].r.score, o1[i].r.score, o0[i].r.nops, o1[i].r.nops, o0[i].r.status, o1[i].r.status);
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
