// lm_merge.hip - the shard merge on the DEVICE: the rows gathered from the ranks already sit in device memory
// (lm_comm.cpp receives them there), so the merging rank orders them where they are and downloads the final order once, instead
// of downloading them, merging on the host threads and copying every 168-byte row twice more (lm_merge_sharded: 0.19 s for the
// 4.7 M rows of a C3 step - a third of what held the 8-GPU model below 0.7).
//
// Order = lm_merge_sharded's = the reference's (lib-index-search.go:2919-2921, merge-search-results.go:142-194): queries ascending
// (batch order); inside a query the subject genomes by the similarity (bitscore * pident, float64) of their best HSP cluster,
// descending, ties by genome key ascending; a genome's rows as its shard emitted them; `hits` = subject genomes of the query over
// all shards (search.go:463,494).  A genome lives in ONE shard and its rows are contiguous there, so a (query, genome) group is
// a run of one rank's block:
//   k_mg_heads   run heads (query or genome changes, or a rank's block starts) + the row's similarity
//   scan         group number of every row
//   k_mg_groups  one thread per head: first row, rows, best similarity -> the group's 24-byte sort key
//   merge sort   of the keys by (query, best descending, genome key)   [rocPRIM, comparison sort: one call for the 128-bit order]
//   k_mg_qruns   groups per query = hits; scan of the group sizes in sorted order = output positions
//   k_mg_emit    one thread per output row: its group by binary search over the positions, the row copied with hits set and
//                the pointer columns cleared (addresses of other processes)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <cstring>

#include <rocprim/rocprim.hpp>

#include "lm_merge.h"

namespace lm {

struct MgKey {
    uint32_t q, g;  // query; group number (payload)
    uint64_t nbest; // ~(order-preserving bits of the best similarity): ascending = similarity descending
    uint64_t bg;
};
struct MgLess {
    __host__ __device__ bool operator()(const MgKey &a, const MgKey &b) const {
        if (a.q != b.q) return a.q < b.q;
        if (a.nbest != b.nbest) return a.nbest < b.nbest;
        return a.bg < b.bg;
    }
};

struct MgToI64 {
    __host__ __device__ int64_t operator()(const uint32_t &x) const { return (int64_t)x; }
};

__device__ __forceinline__ uint64_t mg_ordered(double x) { // total order of float64 as unsigned (x >= 0 here, any x works)
    uint64_t b = (uint64_t)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

__global__ void k_mg_heads(const lm_hsp *__restrict__ rows, int64_t n, const int64_t *__restrict__ rank_off, int nranks, uint32_t *__restrict__ head,
                           double *__restrict__ sim) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool h = i == 0 || rows[i].query != rows[i - 1].query || rows[i].batch_genome != rows[i - 1].batch_genome;
    for (int r = 1; r < nranks && !h; r++) h = i == rank_off[r];
    head[i] = h ? 1u : 0u;
    sim[i] = (double)rows[i].bitscore * rows[i].pident; // SimilarityScore (lib-index-search.go:2352,2621)
}
__global__ void k_mg_groups(const lm_hsp *__restrict__ rows, int64_t n, const uint32_t *__restrict__ head, const uint32_t *__restrict__ gid,
                            const double *__restrict__ sim, MgKey *__restrict__ keys, uint32_t *__restrict__ first, uint32_t *__restrict__ cnt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !head[i]) return;
    const uint32_t g = gid[i] - 1u;
    double best = 0.0; // (lm_merge_sharded starts at 0 as well: similarities are never negative)
    int64_t j = i;
    do {
        best = sim[j] > best ? sim[j] : best;
        j++;
    } while (j < n && !head[j]);
    first[g] = (uint32_t)i;
    cnt[g] = (uint32_t)(j - i);
    MgKey k;
    k.q = rows[i].query;
    k.g = g;
    k.nbest = ~mg_ordered(best);
    k.bg = rows[i].batch_genome;
    keys[g] = k;
}
// sorted order: the size of every group (for the scan) and, per run of one query, the number of groups = hits
__global__ void k_mg_qruns(const MgKey *__restrict__ keys, int64_t ng, const uint32_t *__restrict__ cnt, uint32_t *__restrict__ size_sorted,
                           uint32_t *__restrict__ hits) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= ng) return;
    size_sorted[p] = cnt[keys[p].g];
    if (p > 0 && keys[p - 1].q == keys[p].q) return; // not the head of its query
    int64_t e = p + 1;
    while (e < ng && keys[e].q == keys[p].q) e++;
    for (int64_t x = p; x < e; x++) hits[x] = (uint32_t)(e - p);
}
__global__ void k_mg_emit(const lm_hsp *__restrict__ rows, int64_t n, const MgKey *__restrict__ keys, int64_t ng, const uint32_t *__restrict__ first,
                          const int64_t *__restrict__ outpos, const uint32_t *__restrict__ hits, lm_hsp *__restrict__ out) {
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n) return;
    int64_t lo = 0, hi = ng; // last p with outpos[p] <= o
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (outpos[mid] <= o)
            lo = mid;
        else
            hi = mid;
    }
    const uint32_t g = keys[lo].g;
    lm_hsp h = rows[(int64_t)first[g] + (o - outpos[lo])];
    h.hits = hits[lo];
    h.genome_id = h.seq_id = nullptr; // addresses of another process: the host re-attaches the names
    h.cigar = h.qseq = h.sseq = h.align = nullptr;
    out[o] = h;
}

#define MG_HIP(expr)                        \
    do {                                    \
        hipError_t e_ = (expr);             \
        if (e_ != hipSuccess) return e_;    \
    } while (0)

void *MergeScratch::take(size_t bytes) {
    if (!borrow || nborrowed >= 16) return nullptr;
    void *p = borrow(ctx, bytes);
    if (p) borrowed[nborrowed++] = p;
    return p;
}
void MergeScratch::end_call() {
    if (!borrow) return;
    for (int i = 0; i < nborrowed; i++) give_back(ctx, borrowed[i]);
    nborrowed = 0;
    void **ps[] = {&head, &gid, &sim, &keys, &keys2, &first, &cnt, &sizes, &hits, &outpos, &tmp, &off};
    for (void **p : ps) *p = nullptr;
    for (size_t &c : cap) c = 0;
    borrow = nullptr;
    give_back = nullptr;
    ctx = nullptr;
}
static hipError_t mg_grow(MergeScratch &S, void **p, size_t *cap, size_t need) {
    if (S.borrow) { // (one exact block per buffer and call; a buffer that must grow within the call - tmp - gets a new one)
        if (need <= *cap && *p) return hipSuccess;
        *p = S.take(need + 256);
        *cap = *p ? need + 256 : 0;
        return *p ? hipSuccess : hipErrorOutOfMemory;
    }
    if (need <= *cap && *p) return hipSuccess;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    const size_t want = need + need / 4 + 4096;
    hipError_t e = hipMalloc(p, want);
    if (e == hipSuccess) *cap = want;
    return e;
}

void MergeScratch::release() {
    if (borrow) {
        end_call();
        return;
    }
    void **ps[] = {&head, &gid, &sim, &keys, &keys2, &first, &cnt, &sizes, &hits, &outpos, &tmp, &off};
    for (void **p : ps) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    for (size_t &c : cap) c = 0;
}

// d_rows: the rows of rank 0, 1, ... back to back (rank r at [off[r], off[r + 1]), each block grouped by query ascending, a
// genome's rows together); d_out: n rows in the final order.  Everything on `st`; returns when the work is QUEUED.
hipError_t merge_rows_device(hipStream_t st, const lm_hsp *d_rows, size_t n, const int64_t *off_host, int nranks, lm_hsp *d_out, MergeScratch &S) {
    if (n == 0) return hipSuccess;
    if (n >= ((size_t)1 << 32)) return hipErrorInvalidValue;
    const int64_t N = (int64_t)n;
    MG_HIP(mg_grow(S, &S.head, &S.cap[0], n * 4));
    MG_HIP(mg_grow(S, &S.gid, &S.cap[1], n * 4));
    MG_HIP(mg_grow(S, &S.sim, &S.cap[2], n * 8));
    MG_HIP(mg_grow(S, &S.off, &S.cap[11], (size_t)(nranks + 1) * 8));
    MG_HIP(hipMemcpyAsync(S.off, off_host, (size_t)(nranks + 1) * 8, hipMemcpyHostToDevice, st));
    const int B = 256;
    const unsigned gb = (unsigned)((n + B - 1) / B);
    hipLaunchKernelGGL(k_mg_heads, dim3(gb), dim3(B), 0, st, d_rows, N, (const int64_t *)S.off, nranks, (uint32_t *)S.head, (double *)S.sim);
    size_t bytes = 0;
    MG_HIP(rocprim::inclusive_scan(nullptr, bytes, (uint32_t *)S.head, (uint32_t *)S.gid, n, rocprim::plus<uint32_t>(), st));
    MG_HIP(mg_grow(S, &S.tmp, &S.cap[10], bytes));
    MG_HIP(rocprim::inclusive_scan(S.tmp, bytes, (uint32_t *)S.head, (uint32_t *)S.gid, n, rocprim::plus<uint32_t>(), st));
    uint32_t ng32 = 0;
    MG_HIP(hipMemcpyAsync(&ng32, (uint32_t *)S.gid + (n - 1), 4, hipMemcpyDeviceToHost, st));
    MG_HIP(hipStreamSynchronize(st)); // (the number of groups sizes what follows)
    const size_t ng = ng32;
    MG_HIP(mg_grow(S, &S.keys, &S.cap[3], ng * sizeof(MgKey)));
    MG_HIP(mg_grow(S, &S.keys2, &S.cap[4], ng * sizeof(MgKey)));
    MG_HIP(mg_grow(S, &S.first, &S.cap[5], ng * 4));
    MG_HIP(mg_grow(S, &S.cnt, &S.cap[6], ng * 4));
    MG_HIP(mg_grow(S, &S.sizes, &S.cap[7], ng * 4));
    MG_HIP(mg_grow(S, &S.hits, &S.cap[8], ng * 4));
    MG_HIP(mg_grow(S, &S.outpos, &S.cap[9], (ng + 1) * 8));
    hipLaunchKernelGGL(k_mg_groups, dim3(gb), dim3(B), 0, st, d_rows, N, (const uint32_t *)S.head, (const uint32_t *)S.gid, (const double *)S.sim,
                       (MgKey *)S.keys, (uint32_t *)S.first, (uint32_t *)S.cnt);
    bytes = 0;
    MG_HIP(rocprim::merge_sort(nullptr, bytes, (MgKey *)S.keys, (MgKey *)S.keys2, ng, MgLess(), st));
    MG_HIP(mg_grow(S, &S.tmp, &S.cap[10], bytes));
    MG_HIP(rocprim::merge_sort(S.tmp, bytes, (MgKey *)S.keys, (MgKey *)S.keys2, ng, MgLess(), st));
    const unsigned gg = (unsigned)((ng + B - 1) / B);
    hipLaunchKernelGGL(k_mg_qruns, dim3(gg), dim3(B), 0, st, (const MgKey *)S.keys2, (int64_t)ng, (const uint32_t *)S.cnt, (uint32_t *)S.sizes,
                       (uint32_t *)S.hits);
    auto it = rocprim::make_transform_iterator((const uint32_t *)S.sizes, MgToI64());
    bytes = 0;
    MG_HIP(rocprim::exclusive_scan(nullptr, bytes, it, (int64_t *)S.outpos, (int64_t)0, ng, rocprim::plus<int64_t>(), st));
    MG_HIP(mg_grow(S, &S.tmp, &S.cap[10], bytes));
    MG_HIP(rocprim::exclusive_scan(S.tmp, bytes, it, (int64_t *)S.outpos, (int64_t)0, ng, rocprim::plus<int64_t>(), st));
    hipLaunchKernelGGL(k_mg_emit, dim3(gb), dim3(B), 0, st, d_rows, N, (const MgKey *)S.keys2, (int64_t)ng, (const uint32_t *)S.first,
                       (const int64_t *)S.outpos, (const uint32_t *)S.hits, d_out);
    return hipGetLastError();
}

} // namespace lm
