// lm_prims.h — rocPRIM device primitives used between the kernels (plumbing: radix sorts, scans, run-length encode).
// Called directly (rocprim::), sizes are size_t: no 2^31 item limit except for the segmented sort (unsigned int).
#pragma once
#include <cstring>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include "lm_internal.h"

namespace lm {

#define RPCHK(expr)                                                                                          \
    do {                                                                                                     \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess)                                                                                \
            throw HipError(std::string(#expr) + ": " + hipGetErrorString(_e) + " at " + __FILE__ + ":" +     \
                           std::to_string(__LINE__));                                                        \
    } while (0)

template <typename K, typename V>
static void prim_sort_pairs(hipStream_t st, DBuf<uint8_t> &tmp, const K *k_in, K *k_out, const V *v_in, V *v_out, size_t n,
                            int begin_bit, int end_bit) {
    if (n == 0) return;
    size_t bytes = 0;
    RPCHK(rocprim::radix_sort_pairs(nullptr, bytes, k_in, k_out, v_in, v_out, n, (unsigned)begin_bit, (unsigned)end_bit, st));
    tmp.ensure(bytes);
    RPCHK(rocprim::radix_sort_pairs(tmp.p, bytes, k_in, k_out, v_in, v_out, n, (unsigned)begin_bit, (unsigned)end_bit, st));
}

template <typename K>
static void prim_sort_keys(hipStream_t st, DBuf<uint8_t> &tmp, const K *k_in, K *k_out, size_t n, int begin_bit, int end_bit) {
    if (n == 0) return;
    size_t bytes = 0;
    RPCHK(rocprim::radix_sort_keys(nullptr, bytes, k_in, k_out, n, (unsigned)begin_bit, (unsigned)end_bit, st));
    tmp.ensure(bytes);
    RPCHK(rocprim::radix_sort_keys(tmp.p, bytes, k_in, k_out, n, (unsigned)begin_bit, (unsigned)end_bit, st));
}

// segments [off[s], off[s+1]) sorted independently by key; n < 2^32
template <typename K, typename V, typename Off>
static void prim_segmented_sort_pairs(hipStream_t st, DBuf<uint8_t> &tmp, const K *k_in, K *k_out, const V *v_in, V *v_out,
                                      size_t n, size_t nseg, const Off *off, int begin_bit, int end_bit) {
    if (n == 0 || nseg == 0) return;
    if (n >= ((size_t)1 << 32) || nseg >= ((size_t)1 << 32)) throw HipError("segmented sort: more than 2^32 items");
    size_t bytes = 0;
    RPCHK(rocprim::segmented_radix_sort_pairs(nullptr, bytes, k_in, k_out, v_in, v_out, (unsigned)n, (unsigned)nseg, off,
                                              off + 1, (unsigned)begin_bit, (unsigned)end_bit, st));
    tmp.ensure(bytes);
    RPCHK(rocprim::segmented_radix_sort_pairs(tmp.p, bytes, k_in, k_out, v_in, v_out, (unsigned)n, (unsigned)nseg, off,
                                              off + 1, (unsigned)begin_bit, (unsigned)end_bit, st));
}

struct PrimToI64 {
    template <typename T> __host__ __device__ int64_t operator()(const T &x) const { return (int64_t)x; }
};
// exclusive scan of n+1 counts (counts[n] must be 0) into int64 offsets: offs[n] = total (stays on the device)
template <typename InT> static void prim_scan_to_i64(hipStream_t st, DBuf<uint8_t> &tmp, const InT *counts, size_t n, int64_t *offs) {
    auto it = rocprim::make_transform_iterator(counts, PrimToI64());
    size_t bytes = 0;
    RPCHK(rocprim::exclusive_scan(nullptr, bytes, it, offs, (int64_t)0, n + 1, rocprim::plus<int64_t>(), st));
    tmp.ensure(bytes);
    RPCHK(rocprim::exclusive_scan(tmp.p, bytes, it, offs, (int64_t)0, n + 1, rocprim::plus<int64_t>(), st));
}

template <typename K, typename C>
static void prim_rle(hipStream_t st, DBuf<uint8_t> &tmp, const K *in, size_t n, K *uniq, C *counts, int32_t *nruns) {
    size_t bytes = 0;
    RPCHK(rocprim::run_length_encode(nullptr, bytes, in, n, uniq, counts, nruns, st));
    tmp.ensure(bytes);
    RPCHK(rocprim::run_length_encode(tmp.p, bytes, in, n, uniq, counts, nruns, st));
}

} // namespace lm
