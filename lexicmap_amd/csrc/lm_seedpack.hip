// lm_seedpack.hip — construction of the packed seed image in HBM (DevIndexView).
//
// Replaces the 16 B/seed RAM form of the reference (kv-reader.go:762-1021: per mask a flat []uint64{kmer, value, ...} plus a
// 4^a-entry first-offset table) by, per list md = (mask, direction):
//     part_tab[md][4^a + 1]   u32 offsets of the anchor partitions (the a bases after the mask's p-base prefix)
//     pk_keys                 bit stream, 2 (K-p-a) bits per seed: the rest of the k-mer, ascending inside a partition
//     pk_vals                 bit stream, gid_bits + pos_bits + 1 bits per seed: local genome | position | strand
// = 9.5 B/seed for K=31, p=7, a=6 and 100k x 3-Mb genomes per GPU (+ 0.66 GB of table for M=20000), so that BASELINE
// configs 3-5 fit the 288 GB of one MI355X per shard.  Seeds that do not start with their mask's prefix (only possible
// for genomes that lack the prefix, i.e. tiny ones) keep the flat 16-byte form in per-list outlier arrays.
//
// Built in two passes over the seeds, shown in any order and any batching (loader: chunk files; synthetic builder:
// genome chunks): count() -> partition sizes, place() -> atomic slot per partition + atomicOr into the zeroed streams,
// finish() -> every partition sorted by key in place (one wavefront per partition, LDS bitonic network; partitions
// shared words are merged with masked atomics).  No copy of the unpacked seeds ever exists.
#include "lm_prims.h"

#include <errno.h>
#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <thread>

namespace lm {

#define SP_MAXN 1024 /* largest partition the LDS sorter takes; beyond: rocPRIM fallback */

struct SpParams {
    const uint64_t *masks;
    int K, p, a, P1, key_bits, gid_bits, pos_bits, M;
    const int64_t *batch_first;
    int nbatches, shard_rank, shard_count;
    const int32_t *g2local;
};

__device__ __forceinline__ int64_t sp_local_genome(const SpParams &sp, uint64_t bg) {
    const uint64_t batch = bg >> 17, gi = bg & 0x1ffff;
    int64_t g = (batch < (uint64_t)sp.nbatches ? sp.batch_first[batch] : 0) + (int64_t)gi;
    if (sp.g2local) return sp.g2local[g];
    if (sp.shard_count > 1) g /= sp.shard_count;
    return g;
}

// returns the table slot (md*P1 + part) of a seed, or -1 - md for an outlier
__device__ __forceinline__ int64_t sp_classify(const SpParams &sp, uint32_t m, uint64_t kmer, uint64_t val) {
    const uint32_t md = (m << 1) | (uint32_t)(val & 1ull);
    const int sh = (sp.K - sp.p) << 1;
    if ((kmer >> sh) != (sp.masks[m] >> sh)) return -1 - (int64_t)md;
    const uint32_t part = (uint32_t)(kmer >> sp.key_bits) & (uint32_t)(sp.P1 - 2);
    return (int64_t)md * sp.P1 + part;
}

__global__ void k_sp_count(SpParams sp, const uint16_t *__restrict__ mask, const uint64_t *__restrict__ kmer,
                           const uint64_t *__restrict__ val, int64_t n, uint32_t *__restrict__ tab,
                           unsigned long long *__restrict__ out_cnt) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = sp_classify(sp, mask[i], kmer[i], val[i]);
        if (s >= 0)
            atomicAdd(&tab[s + 1], 1u); // count of partition `part` sits one entry up: see k_sp_scan_rows
        else
            atomicAdd(&out_cnt[-1 - s], 1ull);
    }
}

// per list: counts in row[1..P] -> row[i+1] = first seed of partition i (exclusive scan), row[0] = 0; total -> md_cnt.
// place() then uses row[i+1] as the cursor of partition i, which leaves row[i+1] = first seed of partition i+1: the
// final table needs no second array.
__global__ __launch_bounds__(256) void k_sp_scan_rows(uint32_t *__restrict__ tab, int P1, int64_t nmd,
                                                      unsigned long long *__restrict__ md_cnt) {
    __shared__ uint32_t part_sum[256];
    const int P = P1 - 1;
    const int ipt = (P + 255) / 256;
    for (int64_t md = blockIdx.x; md < nmd; md += gridDim.x) {
        uint32_t *row = tab + md * P1 + 1;
        const int b = threadIdx.x * ipt, e = min(P, b + ipt);
        uint32_t s = 0;
        for (int i = b; i < e; i++) s += row[i];
        part_sum[threadIdx.x] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t acc = 0;
            for (int t = 0; t < 256; t++) {
                const uint32_t v = part_sum[t];
                part_sum[t] = acc;
                acc += v;
            }
            md_cnt[md] = acc;
        }
        __syncthreads();
        uint32_t acc = part_sum[threadIdx.x];
        for (int i = b; i < e; i++) {
            const uint32_t v = row[i];
            row[i] = acc;
            acc += v;
        }
        __syncthreads();
    }
}

__device__ __forceinline__ void sp_bits_or(uint64_t *a, int64_t i, int w, uint64_t v) {
    const int64_t bit = i * (int64_t)w;
    const int64_t word = bit >> 6;
    const int sh = (int)(bit & 63);
    atomicOr((unsigned long long *)&a[word], (unsigned long long)(v << sh));
    if (sh + w > 64) atomicOr((unsigned long long *)&a[word + 1], (unsigned long long)(v >> (64 - sh)));
}

__global__ void k_sp_place(SpParams sp, const uint16_t *__restrict__ mask, const uint64_t *__restrict__ kmer,
                           const uint64_t *__restrict__ val, int64_t n, uint32_t *__restrict__ tab,
                           const int64_t *__restrict__ md_off, uint64_t *__restrict__ pk_keys, uint64_t *__restrict__ pk_vals,
                           const int64_t *__restrict__ out_off, unsigned long long *__restrict__ out_cur,
                           uint64_t *__restrict__ out_kmers, uint64_t *__restrict__ out_vals) {
    const uint64_t km = (1ull << sp.key_bits) - 1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t k = kmer[i], v = val[i];
        const int64_t s = sp_classify(sp, mask[i], k, v);
        if (s >= 0) {
            const int64_t md = s / sp.P1;
            const int64_t slot = md_off[md] + (int64_t)atomicAdd(&tab[s + 1], 1u);
            sp_bits_or(pk_keys, slot, sp.key_bits, k & km);
            sp_bits_or(pk_vals, slot, sp.gid_bits + sp.pos_bits + 1,
                       lm_pack_seed_val((uint64_t)sp_local_genome(sp, v >> 30), v, sp.pos_bits));
        } else {
            const int64_t md = -1 - s;
            const int64_t o = out_off[md] + (int64_t)atomicAdd(&out_cur[md], 1ull);
            out_kmers[o] = k;
            out_vals[o] = v;
        }
    }
}

// ---- in-place sort of the partitions -----------------------------------------------------------------------------------
template <typename Get>
__device__ __forceinline__ void sp_store_range(uint64_t *a, int64_t first, int64_t n, int width, Get get, int lane, int nl) {
    const int64_t w0 = (first * width) >> 6, w1 = ((first + n) * width - 1) >> 6;
    for (int64_t w = w0 + lane; w <= w1; w += nl) {
        uint64_t m;
        const uint64_t v = lm_bits_build_word(w, first, n, width, get, &m);
        if (m == ~0ull) {
            a[w] = v;
        } else { // a word shared with the neighbouring partitions: only our bits
            atomicAnd((unsigned long long *)&a[w], (unsigned long long)~m);
            atomicOr((unsigned long long *)&a[w], (unsigned long long)v);
        }
    }
}

// one wavefront per partition; partitions above SP_MAXN go to `big` (list of table slots)
__global__ __launch_bounds__(64) void k_sp_sort_parts(const uint32_t *__restrict__ tab, const int64_t *__restrict__ md_off,
                                                      int P1, int64_t nmd, int key_bits, int val_bits,
                                                      uint64_t *__restrict__ pk_keys, uint64_t *__restrict__ pk_vals,
                                                      unsigned long long *__restrict__ big, unsigned long long big_cap,
                                                      unsigned long long *__restrict__ nbig, int dbg_mode) {
    __shared__ uint64_t sk[SP_MAXN], sv[SP_MAXN];
    const int lane = threadIdx.x;
    const int P = P1 - 1;
    const int64_t nparts = nmd * P;
    for (int64_t pi = blockIdx.x; pi < nparts; pi += gridDim.x) {
        const int64_t md = pi / P, part = pi % P;
        const uint32_t *row = tab + md * P1 + part;
        const int64_t n = (int64_t)row[1] - (int64_t)row[0];
        if (n <= 1) continue;
        const int64_t first = md_off[md] + row[0];
        if (n > SP_MAXN) {
            if (lane == 0) {
                const unsigned long long o = atomicAdd(nbig, 1ull);
                if (o < big_cap) big[o] = (unsigned long long)(md * P1 + part);
            }
            continue;
        }
        int npad = 2;
        while (npad < n) npad <<= 1;
        bool sorted = true;
        for (int i = lane; i < npad; i += 64) {
            if (i < n) {
                sk[i] = lm_bits_get(pk_keys, first + i, key_bits);
                sv[i] = lm_bits_get(pk_vals, first + i, val_bits);
            } else {
                sk[i] = ~0ull;
                sv[i] = ~0ull;
            }
        }
        __syncthreads();
        for (int i = lane; i + 1 < n; i += 64)
            if (sk[i] > sk[i + 1]) sorted = false;
        if (__ballot(!sorted) == 0) { // nothing to do (chunks arrive in key order more often than not)
            __syncthreads();
            continue;
        }
        if (dbg_mode & 1) {
            __syncthreads();
            continue;
        }
        for (int k = 2; k <= ((dbg_mode & 2) ? 1 : npad); k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = lane; i < (npad >> 1); i += 64) {
                    const int x = ((i & ~(j - 1)) << 1) | (i & (j - 1)), y = x | j;
                    const bool up = (x & k) == 0;
                    const uint64_t kx = sk[x], ky = sk[y];
                    if ((kx > ky) == up && kx != ky) {
                        sk[x] = ky;
                        sk[y] = kx;
                        const uint64_t t = sv[x];
                        sv[x] = sv[y];
                        sv[y] = t;
                    }
                }
                __syncthreads();
            }
        }
        if (!(dbg_mode & 4)) {
            sp_store_range(pk_keys, first, n, key_bits, [&](int64_t i) { return sk[i]; }, lane, 64);
            sp_store_range(pk_vals, first, n, val_bits, [&](int64_t i) { return sv[i]; }, lane, 64);
        }
        __syncthreads();
    }
}

// ---- partitions above SP_MAXN seeds ------------------------------------------------------------------------------------
// Every mask has a few of them: the k-mers a mask captures are the ones closest to it, so the captures of ALL genomes under a
// mask share ~log4(genome length) bases with the mask and pile up in the partitions around the mask's own 13-base prefix
// (thousands to 10^5 seeds at 10^4-10^5 genomes).  They are unpacked in batches, sorted by one rocPRIM segmented radix sort
// per batch and packed back.
__global__ void k_sp_big_info(const unsigned long long *__restrict__ slots, int64_t nb, const uint32_t *__restrict__ tab,
                              const int64_t *__restrict__ md_off, int P1, int64_t *__restrict__ first, int64_t *__restrict__ cnt) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long s = slots[i];
        first[i] = md_off[s / P1] + tab[s];
        cnt[i] = (int64_t)tab[s + 1] - (int64_t)tab[s];
    }
}
// element e of the batch = seed (e - off[p]) of big partition p (off relative to the batch)
__global__ void k_sp_unpack_many(const uint64_t *__restrict__ keys, const uint64_t *__restrict__ vals, int key_bits, int val_bits,
                                 const int64_t *__restrict__ first, const int64_t *__restrict__ off, int64_t nseg, int64_t base,
                                 int64_t total, uint64_t *__restrict__ ko, uint64_t *__restrict__ vo) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t lo = 0, hi = nseg; // last p with off[p] - base <= e
        while (lo + 1 < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (off[mid] - base <= e)
                lo = mid;
            else
                hi = mid;
        }
        const int64_t src = first[lo] + (e - (off[lo] - base));
        ko[e] = lm_bits_get(keys, src, key_bits);
        vo[e] = lm_bits_get(vals, src, val_bits);
    }
}
__global__ void k_sp_rel_offsets(const int64_t *__restrict__ off, int64_t nseg, int64_t base, uint32_t *__restrict__ rel) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= nseg; i += (int64_t)gridDim.x * blockDim.x)
        rel[i] = (uint32_t)(off[i] - base);
}
__global__ __launch_bounds__(256) void k_sp_repack_many(uint64_t *__restrict__ keys, uint64_t *__restrict__ vals, int key_bits,
                                                         int val_bits, const int64_t *__restrict__ first,
                                                         const uint32_t *__restrict__ rel, int64_t nseg,
                                                         const uint64_t *__restrict__ ki, const uint64_t *__restrict__ vi) {
    for (int64_t p = blockIdx.x; p < nseg; p += gridDim.x) {
        const int64_t n = (int64_t)rel[p + 1] - (int64_t)rel[p];
        const uint64_t *k = ki + rel[p], *v = vi + rel[p];
        sp_store_range(keys, first[p], n, key_bits, [&](int64_t i) { return k[i]; }, (int)threadIdx.x, (int)blockDim.x);
        sp_store_range(vals, first[p], n, val_bits, [&](int64_t i) { return v[i]; }, (int)threadIdx.x, (int)blockDim.x);
    }
}

// all seeds of one list back in the reference's (k-mer, value) form
__global__ void k_sp_dump_list(DevIndexView ix, uint32_t md, int64_t n_main, uint64_t *__restrict__ kmers,
                               uint64_t *__restrict__ vals) {
    const int64_t base = ix.md_off[md];
    const uint32_t *row = ix.part_tab + (int64_t)md * ix.P1;
    const uint64_t pfx = ix.masks[md >> 1] >> ((ix.K - ix.mask_prefix) << 1);
    const int P = ix.P1 - 1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_main; i += (int64_t)gridDim.x * blockDim.x) {
        int lo = 0, hi = P; // partition of seed i: last row entry <= i
        while (lo + 1 < hi) {
            const int mid = (lo + hi) >> 1;
            if ((int64_t)row[mid] <= i)
                lo = mid;
            else
                hi = mid;
        }
        const uint64_t rem = lm_bits_get(ix.pk_keys, base + i, ix.key_bits);
        const uint64_t pv = lm_bits_get(ix.pk_vals, base + i, ix.gid_bits + ix.pos_bits + 1);
        kmers[i] = (((pfx << (ix.part_bases << 1)) | (uint64_t)lo) << ix.key_bits) | rem;
        vals[i] = lm_unpack_seed_val(pv, ix.g_bg[lm_packed_val_genome(pv, ix.pos_bits)], ix.pos_bits, (int)(md & 1));
    }
}

static int sp_grid(int64_t n, int block = 256) {
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > 262144) g = 262144;
    return (int)g;
}

static SpParams sp_params(const SeedPacker &sp) {
    const lm_index *ix = sp.ix;
    SpParams p;
    p.masks = ix->view.masks;
    p.K = ix->host.k;
    p.p = ix->host.mask_prefix;
    p.a = sp.a;
    p.P1 = sp.P1;
    p.key_bits = sp.key_bits;
    p.gid_bits = sp.gid_bits;
    p.pos_bits = sp.pos_bits;
    p.M = ix->host.M;
    p.batch_first = ix->view.batch_first;
    p.nbatches = ix->view.nbatches;
    p.shard_rank = ix->view.shard_rank;
    p.shard_count = ix->view.shard_count;
    p.g2local = ix->view.g2local;
    return p;
}

static int bits_for(int64_t n) { // bits needed for values in [0, n)
    int b = 1;
    while (((int64_t)1 << b) < n) b++;
    return b;
}

void SeedPacker::begin(lm_index *ix_, int64_t local_genomes, int64_t max_genome_len) {
    ix = ix_;
    const HostIndex &h = ix->host;
    if (h.M > 65535) throw HipError("seed image: more than 65535 masks are not supported (16-bit mask ids in the loader's staging records; include/lexicmap_hip.h)");
    a = std::min(h.anchor_prefix, 6);
    while (a > 0 && h.mask_prefix + a >= h.k) a--;
    P1 = (1 << (2 * a)) + 1;
    key_bits = 2 * (h.k - h.mask_prefix - a);
    gid_bits = bits_for(std::max<int64_t>(local_genomes, 2));
    pos_bits = bits_for(std::max<int64_t>(max_genome_len + 1, 2));
    if (pos_bits > 28) throw HipError("seed image: genome longer than 2^28 bases (lib-index-build.go:421-425)");
    if (gid_bits + pos_bits + 1 > 64) throw HipError("seed image: value does not fit 64 bits");
    const int64_t nmd = 2ll * h.M;
    ix->d_part_tab.alloc_exact((size_t)(nmd * P1), true, ix->st);
    out_cnt.alloc_exact((size_t)nmd + 1, true, ix->st);
    n_main = n_out = 0;
    placing = false;
}

void SeedPacker::count(const uint16_t *mask, const uint64_t *kmer, const uint64_t *val, int64_t n) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_sp_count, dim3(sp_grid(n)), dim3(256), 0, ix->st, sp_params(*this), mask, kmer, val, n,
                       ix->d_part_tab.p, out_cnt.p);
}

void SeedPacker::end_count() {
    const int64_t nmd = 2ll * ix->host.M;
    DBuf<unsigned long long> md_cnt;
    md_cnt.alloc_exact((size_t)nmd + 1, true, ix->st);
    hipLaunchKernelGGL(k_sp_scan_rows, dim3((unsigned)std::min<int64_t>(nmd, 65536)), dim3(256), 0, ix->st, ix->d_part_tab.p,
                       P1, nmd, md_cnt.p);
    ix->d_md_off.alloc_exact((size_t)nmd + 1);
    ix->d_out_off.alloc_exact((size_t)nmd + 1);
    prim_scan_to_i64(ix->st, ix->tmp, md_cnt.p, (size_t)nmd, ix->d_md_off.p);
    prim_scan_to_i64(ix->st, ix->tmp, out_cnt.p, (size_t)nmd, ix->d_out_off.p);
    int64_t tot[2];
    HIPCHK(hipMemcpyAsync(&tot[0], ix->d_md_off.p + nmd, sizeof(int64_t), hipMemcpyDeviceToHost, ix->st));
    HIPCHK(hipMemcpyAsync(&tot[1], ix->d_out_off.p + nmd, sizeof(int64_t), hipMemcpyDeviceToHost, ix->st));
    HIPCHK(hipStreamSynchronize(ix->st));
    n_main = tot[0];
    n_out = tot[1];
    const int val_bits = gid_bits + pos_bits + 1;
    // +2 words: lm_bits_get may touch the word after the last element
    ix->d_pk_keys.alloc_exact((size_t)((n_main * key_bits + 63) / 64 + 2), true, ix->st);
    ix->d_pk_vals.alloc_exact((size_t)((n_main * val_bits + 63) / 64 + 2), true, ix->st);
    ix->d_out_kmers.alloc_exact((size_t)n_out + 1);
    ix->d_out_vals.alloc_exact((size_t)n_out + 1);
    HIPCHK(hipMemsetAsync(out_cnt.p, 0, ((size_t)nmd + 1) * sizeof(unsigned long long), ix->st)); // now the outlier cursors
    placing = true;
}

void SeedPacker::place(const uint16_t *mask, const uint64_t *kmer, const uint64_t *val, int64_t n) {
    if (n <= 0) return;
    if (!placing) throw HipError("SeedPacker::place before end_count");
    hipLaunchKernelGGL(k_sp_place, dim3(sp_grid(n)), dim3(256), 0, ix->st, sp_params(*this), mask, kmer, val, n,
                       ix->d_part_tab.p, ix->d_md_off.p, ix->d_pk_keys.p, ix->d_pk_vals.p, ix->d_out_off.p, out_cnt.p,
                       ix->d_out_kmers.p, ix->d_out_vals.p);
}

void SeedPacker::finish() {
    const HostIndex &h = ix->host;
    const int64_t nmd = 2ll * h.M;
    const int val_bits = gid_bits + pos_bits + 1;
    // ---- main partitions
    // (a partition counts as big above SP_MAXN seeds, so there are fewer than n_main / SP_MAXN of them: 6e5 at C3, 1.5e6 in a
    // 250 000-genome shard of C4)
    const unsigned long long big_cap = (unsigned long long)(n_main / SP_MAXN) + 1024;
    DBuf<unsigned long long> big, nbig;
    big.alloc_exact(big_cap);
    nbig.alloc_exact(1, true, ix->st);
    if (n_main > 0) {
        const int64_t nparts = nmd * (P1 - 1);
        const int dbg_mode = 0; // (the kernel's timing-experiment modes are not reachable any more)
        const double t0 = now_ms();
        hipLaunchKernelGGL(k_sp_sort_parts, dim3((unsigned)std::min<int64_t>(nparts, (int64_t)1 << 22)), dim3(64), 0, ix->st,
                           ix->d_part_tab.p, ix->d_md_off.p, P1, nmd, key_bits, val_bits, ix->d_pk_keys.p, ix->d_pk_vals.p,
                           big.p, big_cap, nbig.p, dbg_mode);
        unsigned long long nb = 0;
        HIPCHK(hipMemcpyAsync(&nb, nbig.p, sizeof nb, hipMemcpyDeviceToHost, ix->st));
        HIPCHK(hipStreamSynchronize(ix->st));
        if (getenv("LM_DEBUG"))
            fprintf(stderr, "[lm] seed image: partition sort kernel %.0f ms (mode %d), %llu partitions above %d seeds\n", now_ms() - t0,
                    dbg_mode, nb, SP_MAXN);
        if (nb > big_cap) throw HipError("seed image: too many partitions above the LDS sorter's size");
        if (nb > 0) {
            const double t1 = now_ms();
            DBuf<int64_t> bfirst, bcnt, boff;
            bfirst.alloc_exact((size_t)nb + 1);
            bcnt.alloc_exact((size_t)nb + 1, true, ix->st);
            boff.alloc_exact((size_t)nb + 2);
            hipLaunchKernelGGL(k_sp_big_info, dim3(sp_grid((int64_t)nb)), dim3(256), 0, ix->st, big.p, (int64_t)nb, ix->d_part_tab.p,
                               ix->d_md_off.p, P1, bfirst.p, bcnt.p);
            prim_scan_to_i64(ix->st, ix->tmp, bcnt.p, (size_t)nb, boff.p);
            std::vector<int64_t> off((size_t)nb + 1);
            HIPCHK(hipMemcpyAsync(off.data(), boff.p, ((size_t)nb + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, ix->st));
            HIPCHK(hipStreamSynchronize(ix->st));
            // batches of whole partitions: 4 temporary u64 arrays, at most a quarter of what is free and 2^31 elements
            size_t fr = 0, tot = 0;
            (void)hipMemGetInfo(&fr, &tot);
            int64_t batch = std::min<int64_t>(((int64_t)1 << 31) - 1, std::max<int64_t>((int64_t)(fr / 4 / 40), 1 << 20));
            DBuf<uint64_t> k0, k1, v0, v1;
            DBuf<uint32_t> rel;
            int64_t done_elems = 0;
            for (size_t b0 = 0; b0 < (size_t)nb;) {
                size_t b1 = b0 + 1;
                while (b1 < (size_t)nb && off[b1 + 1] - off[b0] <= batch) b1++;
                const int64_t E = off[b1] - off[b0], nseg = (int64_t)(b1 - b0);
                if (E >= (int64_t)1 << 32) throw HipError("seed image: one partition holds more than 2^32 seeds");
                k0.ensure((size_t)E);
                k1.ensure((size_t)E);
                v0.ensure((size_t)E);
                v1.ensure((size_t)E);
                rel.ensure((size_t)nseg + 1);
                hipLaunchKernelGGL(k_sp_unpack_many, dim3(sp_grid(E)), dim3(256), 0, ix->st, ix->d_pk_keys.p, ix->d_pk_vals.p, key_bits,
                                   val_bits, bfirst.p + b0, boff.p + b0, nseg, off[b0], E, k0.p, v0.p);
                hipLaunchKernelGGL(k_sp_rel_offsets, dim3(sp_grid(nseg + 1)), dim3(256), 0, ix->st, boff.p + b0, nseg, off[b0], rel.p);
                prim_segmented_sort_pairs(ix->st, ix->tmp, k0.p, k1.p, v0.p, v1.p, (size_t)E, (size_t)nseg, rel.p, 0, key_bits);
                hipLaunchKernelGGL(k_sp_repack_many, dim3((unsigned)std::min<int64_t>(nseg, 65536)), dim3(256), 0, ix->st,
                                   ix->d_pk_keys.p, ix->d_pk_vals.p, key_bits, val_bits, bfirst.p + b0, rel.p, nseg, k1.p, v1.p);
                HIPCHK(hipStreamSynchronize(ix->st));
                done_elems += E;
                b0 = b1;
            }
            if (getenv("LM_DEBUG"))
                fprintf(stderr, "[lm] seed image: %llu large partitions (%lld seeds) sorted by rocPRIM in %.0f ms\n", nb,
                        (long long)done_elems, now_ms() - t1);
        }
    }
    // ---- outlier lists: per list by k-mer
    if (n_out > 0) {
        DBuf<uint64_t> k1, v1;
        k1.alloc_exact((size_t)n_out + 1);
        v1.alloc_exact((size_t)n_out + 1);
        prim_segmented_sort_pairs(ix->st, ix->tmp, ix->d_out_kmers.p, k1.p, ix->d_out_vals.p, v1.p, (size_t)n_out, (size_t)nmd,
                                  ix->d_out_off.p, 0, 2 * h.k);
        HIPCHK(hipStreamSynchronize(ix->st));
        std::swap(ix->d_out_kmers.p, k1.p);
        std::swap(ix->d_out_kmers.cap, k1.cap);
        std::swap(ix->d_out_vals.p, v1.p);
        std::swap(ix->d_out_vals.cap, v1.cap);
    }
    HIPCHK(hipStreamSynchronize(ix->st));
    out_cnt.release();
    DevIndexView &v = ix->view;
    v.part_bases = a;
    v.P1 = P1;
    v.key_bits = key_bits;
    v.gid_bits = gid_bits;
    v.pos_bits = pos_bits;
    v.pk_keys = ix->d_pk_keys.p;
    v.pk_vals = ix->d_pk_vals.p;
    v.part_tab = ix->d_part_tab.p;
    v.md_off = ix->d_md_off.p;
    v.out_kmers = ix->d_out_kmers.p;
    v.out_vals = ix->d_out_vals.p;
    v.out_off = ix->d_out_off.p;
    ix->n_seeds = n_main + n_out;
    ix->n_seeds_outlier = n_out;
    ix->seed_bytes = (int64_t)(ix->d_pk_keys.cap * 8 + ix->d_pk_vals.cap * 8 + ix->d_part_tab.cap * 4 + ix->d_md_off.cap * 8 +
                               ix->d_out_off.cap * 8 + ix->d_out_kmers.cap * 8 + ix->d_out_vals.cap * 8);
}

} // namespace lm

// (k-mer, value) pairs stored under one mask, both directions, in the reference's in-memory form: what `lexicmap utils
// kmers -m <mask>` lists (cmd/kmers.go:101-180) and what kv.Reader.ReadDataOfAMaskAsList returns (kv-reader.go:762).
// Inspection / test entry point: the search path never unpacks lists.
extern "C" lm_status lm_index_mask_seeds(lm_index *ix, int32_t mask, uint64_t *kmers, uint64_t *vals, size_t cap, size_t *n_out) {
    if (!ix || mask < 0 || mask >= ix->host.M || !n_out) return LM_ERR_ARG;
    try {
        std::lock_guard<std::mutex> lock(ix->mu);
        HIPCHK(hipSetDevice(ix->device));
        const int64_t nmd = 2ll * ix->host.M;
        int64_t mdo[3], oo[3];
        HIPCHK(hipMemcpy(mdo, ix->d_md_off.p + 2 * mask, sizeof mdo, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(oo, ix->d_out_off.p + 2 * mask, sizeof oo, hipMemcpyDeviceToHost));
        (void)nmd;
        const size_t total = (size_t)(mdo[2] - mdo[0]) + (size_t)(oo[2] - oo[0]);
        *n_out = total;
        if (cap == 0 || (!kmers && !vals)) return LM_OK; // size query
        if (!kmers || !vals || cap < total) {
            ix->err = "lm_index_mask_seeds: buffer too small (cap < *n) or a NULL array";
            return LM_ERR_ARG; // nothing was written; *n holds the count to allocate
        }
        DBuf<uint64_t> dk, dv;
        size_t w = 0;
        for (int dir = 0; dir < 2; dir++) {
            const int64_t nm = mdo[dir + 1] - mdo[dir], no = oo[dir + 1] - oo[dir];
            if (nm > 0) {
                dk.ensure((size_t)nm);
                dv.ensure((size_t)nm);
                hipLaunchKernelGGL(lm::k_sp_dump_list, dim3(lm::sp_grid(nm)), dim3(256), 0, ix->st, ix->view,
                                   (uint32_t)(2 * mask + dir), nm, dk.p, dv.p);
                HIPCHK(hipMemcpyAsync(kmers + w, dk.p, (size_t)nm * 8, hipMemcpyDeviceToHost, ix->st));
                HIPCHK(hipMemcpyAsync(vals + w, dv.p, (size_t)nm * 8, hipMemcpyDeviceToHost, ix->st));
                HIPCHK(hipStreamSynchronize(ix->st));
                w += (size_t)nm;
            }
            if (no > 0) {
                HIPCHK(hipMemcpy(kmers + w, ix->d_out_kmers.p + oo[dir], (size_t)no * 8, hipMemcpyDeviceToHost));
                HIPCHK(hipMemcpy(vals + w, ix->d_out_vals.p + oo[dir], (size_t)no * 8, hipMemcpyDeviceToHost));
                w += (size_t)no;
            }
        }
        return LM_OK;
    } catch (const std::exception &e) {
        ix->err = e.what();
        return LM_ERR_HIP;
    }
}

// ---- the HBM image written back to disk -----------------------------------------------------------------------------------
// `lexicmap index` output as the reference lays it out, EXCEPT masks.bin, which is written in this
// build's own LMMASKS1 layout (lexichash's file layout is not in the reference tree): info.toml (lib-index-build.go:1914-1932),
// seeds/chunk_NNN.bin + .idx (kv/kv-data.go:126-602: per mask the distinct k-mers ascending, two per record,
// k-mer deltas and value counts group-varint coded, 7-byte values while there are <= 512 genome batches; the .idx holds the
// first k-mer and offset of every anchor partition present), genomes/batch_NNNN/genomes.bin + .idx (genome/genome.go:217-357),
// genomes.map.bin (lib-index-build.go:649-655).  What it is for here: the GPU-built benchmark sets become reference-format
// indexes, so the loader (lm_index_open) is exercised and timed at their size and its packed image can be compared with the
// one the builder made in HBM.  (The Go `lexicmap search` could read everything but that masks.bin.)
namespace lm {
namespace {
struct OutFile {
    FILE *f = nullptr;
    int64_t n = 0;
    bool bad = false;
    explicit OutFile(const std::string &p) {
        f = fopen(p.c_str(), "wb");
        bad = f == nullptr;
    }
    ~OutFile() {
        if (f) fclose(f);
    }
    void put(const void *p, size_t len) {
        if (!bad && len && fwrite(p, 1, len, f) != len) bad = true;
        n += (int64_t)len;
    }
    void be(uint64_t v, int bytes) {
        uint8_t b[8];
        for (int i = 0; i < bytes; i++) b[i] = (uint8_t)(v >> (8 * (bytes - 1 - i)));
        put(b, (size_t)bytes);
    }
};
inline void push_be(std::vector<uint8_t> &o, uint64_t v, int bytes) {
    for (int i = bytes - 1; i >= 0; i--) o.push_back((uint8_t)(v >> (8 * i)));
}
inline int byte_len(uint64_t v) {
    int n = 1;
    while (n < 8 && (v >> (8 * n)) != 0) n++;
    return n;
}
// control byte + the two numbers in their minimal big-endian widths (util/varint-GB.go:28-46)
inline void push_pair(std::vector<uint8_t> &o, uint64_t a, uint64_t b, uint8_t flags) {
    const int la = byte_len(a), lb = byte_len(b);
    o.push_back((uint8_t)(flags | ((la - 1) << 3) | (lb - 1)));
    push_be(o, a, la);
    push_be(o, b, lb);
}
bool make_dir(const std::string &p) { return mkdir(p.c_str(), 0777) == 0 || errno == EEXIST; }

// one mask's seeds (any order) -> its block of the chunk file (relative offsets) + its .idx records (kmer, rel offset<<1|second)
void encode_mask(std::vector<std::pair<uint64_t, uint64_t>> &kv, int K, int mask_prefix, int anchor_prefix, bool use7,
                 std::vector<uint8_t> &data, std::vector<std::pair<uint64_t, uint64_t>> &idx) {
    data.clear();
    idx.clear();
    std::sort(kv.begin(), kv.end());
    // distinct k-mers: [start, end) runs
    std::vector<size_t> runs;
    for (size_t i = 0; i < kv.size(); i++)
        if (i == 0 || kv[i].first != kv[i - 1].first) runs.push_back(i);
    const size_t nk = runs.size();
    push_be(data, (uint64_t)nk, 8);
    if (nk == 0) return;
    runs.push_back(kv.size());
    const int vb = use7 ? 7 : 8;
    const int shift = (K - mask_prefix - anchor_prefix) << 1;
    const uint64_t amask = ((uint64_t)1 << (anchor_prefix << 1)) - 1;
    uint64_t seen_part = ~(uint64_t)0, prev_second = 0; // previous record's second k-mer (the delta base)
    auto note = [&](uint64_t kmer, bool second, size_t at) { // first k-mer of every anchor partition, in file order
        const uint64_t part = (kmer >> shift) & amask;
        if (part != seen_part) {
            seen_part = part;
            idx.emplace_back(kmer, ((uint64_t)at << 1) | (second ? 1u : 0u));
        }
    };
    for (size_t r = 0; r < nk; r += 2) {
        const bool pair = r + 1 < nk, last = r + 2 >= nk;
        const uint64_t k1 = kv[runs[r]].first, k2 = pair ? kv[runs[r + 1]].first : 0;
        const uint64_t n1 = runs[r + 1] - runs[r], n2 = pair ? runs[r + 2] - runs[r + 1] : 0;
        const size_t at = data.size();
        note(k1, false, at);
        if (pair) note(k2, true, at);
        push_pair(data, k1 - prev_second, pair ? k2 - k1 : 0, (uint8_t)((last ? 128 : 0) | (pair ? 0 : 64)));
        push_pair(data, n1, n2, 0);
        for (size_t i = runs[r]; i < runs[r] + n1 + n2; i++) push_be(data, kv[i].second, vb);
        prev_second = pair ? k2 : k1;
    }
}
} // namespace
} // namespace lm

extern "C" lm_status lm_index_save(lm_index *ix, const char *dir_c, int chunks) {
    using namespace lm;
    if (!ix || !dir_c) return LM_ERR_ARG;
    std::lock_guard<std::mutex> lock(ix->mu);
    const HostIndex &h = ix->host;
    if (h.shard_count > 1) {
        ix->err = "lm_index_save: a shard of an index cannot be saved (open or build it unsharded)";
        return LM_ERR_ARG;
    }
    try {
        HIPCHK(hipSetDevice(ix->device));
        const std::string dir(dir_c);
        const int M = h.M, K = h.k;
        if (chunks < 1) chunks = 1;
        if (chunks > M) chunks = M;
        // info.toml names the number of chunk FILES: ceil(M / ceil(M / chunks)) of them are written (M = 100 masks asked into 16
        // chunks are 15 files of 7 masks), as the reference's writer and the oracle's do
        chunks = (M + (M + chunks - 1) / chunks - 1) / ((M + chunks - 1) / chunks);
        const int nbatches = std::max(1, h.genome_batches > 0 ? h.genome_batches : (int)((h.genomes.size() + 4999) / 5000));
        const bool use7 = nbatches <= 512; // kv-data.go:137
        if (!make_dir(dir) || !make_dir(dir + "/seeds") || !make_dir(dir + "/genomes")) throw HipError("lm_index_save: cannot create " + dir);
        // ---- masks.bin (this build's layout, lm_format.cpp) and info.toml
        {
            OutFile f(dir + "/masks.bin");
            f.put("LMMASKS1", 8);
            f.be((uint64_t)K, 1);
            f.be(0, 3);
            f.be((uint64_t)M, 4);
            f.be(0, 8);
            for (int i = 0; i < M; i++) f.be(h.masks[i], 8);
            if (f.bad) throw HipError("lm_index_save: write failed (masks.bin)");
        }
        int64_t gbases = 0;
        for (auto &g : h.genomes) gbases += g.genome_size;
        {
            OutFile f(dir + "/info.toml");
            char t[1024];
            const int n = snprintf(t, sizeof t,
                                   "# Index format\nmain-version = 3\nminor-version = 5\n# LexicHash\nmax-K = %d\nmasks = %d\nrand-seed = 1\n"
                                   "# Seed distance\nmax-seed-dist = 100\nseed-dist-in-desert = 50\n# Seeds (k-mer-value data) files\nchunks = %d\n"
                                   "index-partitions = %d\n# Input genomes\ninput-genomes = %lld\ninput-bases = %lld\n# Genome data\ngenomes = %lld\n"
                                   "genome-batch-size = 5000\ngenome-batches = %d\ncontig-interval = %d\n",
                                   K, M, chunks, 1 << (2 * h.anchor_prefix), (long long)h.genomes.size(),
                                   (long long)(h.total_bases > 0 ? h.total_bases : gbases), (long long)h.genomes.size(), nbatches,
                                   h.contig_interval);
            f.put(t, (size_t)n);
            if (f.bad) throw HipError("lm_index_save: write failed (info.toml)");
        }
        // ---- genomes: batches of 5000 in dense order, bases straight from the 2-bit store in HBM
        {
            OutFile fmap(dir + "/genomes.map.bin");
            std::vector<uint8_t> bits;
            for (int b = 0; b < nbatches; b++) {
                char nm[64];
                snprintf(nm, sizeof nm, "/genomes/batch_%04d", b);
                if (!make_dir(dir + nm)) throw HipError("lm_index_save: cannot create the genome batch directory");
                OutFile fg(dir + nm + "/genomes.bin"), fi(dir + nm + "/genomes.bin.idx");
                fg.put(".genomes", 8);
                fg.be(0, 1);
                fg.be(1, 1);
                fg.be(0, 6);
                std::vector<const HostGenome *> members;
                for (auto &g : h.genomes)
                    if ((int)(g.bg >> 17) == b) members.push_back(&g);
                std::sort(members.begin(), members.end(), [](const HostGenome *x, const HostGenome *y) { return x->bg < y->bg; });
                fi.put(".genomei", 8);
                fi.be(0, 1);
                fi.be(1, 1);
                fi.be(0, 6);
                fi.be((uint64_t)b, 4);
                fi.be((uint64_t)members.size(), 4);
                for (const HostGenome *g : members) {
                    fi.be((uint64_t)fg.n, 8);
                    fi.be((uint64_t)g->len, 4);
                    fg.be(g->id.size(), 2);
                    fg.put(g->id.data(), g->id.size());
                    fg.be((uint64_t)g->genome_size, 4);
                    fg.be((uint64_t)g->len, 4);
                    fg.be((uint64_t)g->nseqs, 4);
                    for (int s = 0; s < g->nseqs; s++) {
                        fg.be((uint64_t)g->seq_sizes[s], 4);
                        fg.be(g->seq_ids[s].size(), 2);
                        fg.put(g->seq_ids[s].data(), g->seq_ids[s].size());
                    }
                    const size_t nb = ((size_t)g->len + 3) >> 2;
                    bits.resize(nb);
                    HIPCHK(hipMemcpy(bits.data(), ix->d_gbits.p + g->bits_off, nb, hipMemcpyDeviceToHost));
                    fg.be(nb, 4);
                    fg.be((uint64_t)g->len, 4);
                    fg.put(bits.data(), nb);
                    fmap.be(g->id.size(), 2);
                    fmap.put(g->id.data(), g->id.size());
                    fmap.be(g->bg, 8);
                }
                if (fg.bad || fi.bad) throw HipError("lm_index_save: write failed (genomes)");
            }
            if (fmap.bad) throw HipError("lm_index_save: write failed (genomes.map.bin)");
        }
        // ---- genomes.chunks.bin (lib-index-build.go:1787-1808): per split genome the number of its chunks and their keys, in
        // chunk order; written (empty when no genome was split) so that the saved index is complete for the reference's reader
        {
            std::map<int, std::vector<std::pair<int, uint64_t>>> lists; // list number -> (chunk index, key)
            for (const auto &kv : h.chunk_of) lists[kv.second.list].emplace_back(kv.second.idx, kv.first);
            OutFile fc(dir + "/genomes.chunks.bin");
            for (auto &li : lists) {
                if (li.second.size() <= 1) continue;
                std::sort(li.second.begin(), li.second.end());
                fc.be((uint64_t)li.second.size(), 8);
                for (const auto &e : li.second) fc.be(e.second, 8);
            }
            if (fc.bad) throw HipError("lm_index_save: write failed (genomes.chunks.bin)");
        }
        // ---- seeds: masks split evenly over the chunk files (lib-index-build.go:1861-1889)
        std::vector<int64_t> md_off((size_t)2 * M + 1), out_off((size_t)2 * M + 1);
        HIPCHK(hipMemcpy(md_off.data(), ix->d_md_off.p, md_off.size() * 8, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(out_off.data(), ix->d_out_off.p, out_off.size() * 8, hipMemcpyDeviceToHost));
        DBuf<uint64_t> dk, dv;
        const int per = (M + chunks - 1) / chunks;
        for (int c = 0, m0 = 0; m0 < M; c++, m0 += per) {
            const int m1 = std::min(M, m0 + per);
            // every list of the chunk's masks unpacked into one buffer, one copy to the host
            const int64_t s0 = md_off[(size_t)2 * m0], s1 = md_off[(size_t)2 * m1], ns = s1 - s0;
            std::vector<uint64_t> hk((size_t)ns), hv((size_t)ns);
            if (ns > 0) {
                dk.ensure((size_t)ns);
                dv.ensure((size_t)ns);
                for (int md = 2 * m0; md < 2 * m1; md++) {
                    const int64_t nm = md_off[(size_t)md + 1] - md_off[(size_t)md];
                    if (nm > 0)
                        hipLaunchKernelGGL(lm::k_sp_dump_list, dim3(lm::sp_grid(nm)), dim3(256), 0, ix->st, ix->view, (uint32_t)md, nm,
                                           dk.p + (md_off[(size_t)md] - s0), dv.p + (md_off[(size_t)md] - s0));
                }
                HIPCHK(hipMemcpyAsync(hk.data(), dk.p, (size_t)ns * 8, hipMemcpyDeviceToHost, ix->st));
                HIPCHK(hipMemcpyAsync(hv.data(), dv.p, (size_t)ns * 8, hipMemcpyDeviceToHost, ix->st));
                HIPCHK(hipStreamSynchronize(ix->st));
            }
            const int64_t o0 = out_off[(size_t)2 * m0], o1 = out_off[(size_t)2 * m1], no = o1 - o0;
            std::vector<uint64_t> ok((size_t)no), ov((size_t)no);
            if (no > 0) {
                HIPCHK(hipMemcpy(ok.data(), ix->d_out_kmers.p + o0, (size_t)no * 8, hipMemcpyDeviceToHost));
                HIPCHK(hipMemcpy(ov.data(), ix->d_out_vals.p + o0, (size_t)no * 8, hipMemcpyDeviceToHost));
            }
            // encode the masks on the host threads, then lay the blocks out and fix the offsets up
            const int nm = m1 - m0;
            std::vector<std::vector<uint8_t>> blocks((size_t)nm);
            std::vector<std::vector<std::pair<uint64_t, uint64_t>>> recs((size_t)nm);
            std::atomic<int> next{0};
            auto work = [&]() {
                std::vector<std::pair<uint64_t, uint64_t>> kv;
                for (int j; (j = next.fetch_add(1)) < nm;) {
                    const int m = m0 + j;
                    kv.clear();
                    for (int64_t i = md_off[(size_t)2 * m] - s0; i < md_off[(size_t)2 * m + 2] - s0; i++) kv.emplace_back(hk[(size_t)i], hv[(size_t)i]);
                    for (int64_t i = out_off[(size_t)2 * m] - o0; i < out_off[(size_t)2 * m + 2] - o0; i++) kv.emplace_back(ok[(size_t)i], ov[(size_t)i]);
                    encode_mask(kv, K, h.mask_prefix, h.anchor_prefix, use7, blocks[(size_t)j], recs[(size_t)j]);
                }
            };
            {
                const int nth = std::max(1, std::min<int>(16, (int)std::thread::hardware_concurrency()));
                std::vector<std::thread> th;
                for (int t = 1; t < nth; t++) th.emplace_back(work);
                work();
                for (auto &t : th) t.join();
            }
            char nmf[64];
            snprintf(nmf, sizeof nmf, "/seeds/chunk_%03d.bin", c);
            OutFile fd(dir + nmf), fx(dir + nmf + ".idx");
            fd.put(".kv-data", 8);
            fd.be(1, 1);
            fd.be(1, 1);
            fd.be((uint64_t)K, 1);
            fd.be(use7 ? 1 : 0, 1);
            fd.be(0, 4);
            fd.be((uint64_t)m0, 8);
            fd.be((uint64_t)nm, 8);
            fx.put(".kvindex", 8);
            fx.be(1, 1);
            fx.be(1, 1);
            fx.be((uint64_t)K, 1);
            fx.be((uint64_t)h.mask_prefix, 1);
            fx.be((uint64_t)h.anchor_prefix, 1);
            fx.be(use7 ? 1 : 0, 1);
            fx.be(0, 2);
            fx.be((uint64_t)m0, 8);
            fx.be((uint64_t)nm, 8);
            for (int j = 0; j < nm; j++) {
                const int64_t at = fd.n; // the mask's block starts with its k-mer count
                fd.put(blocks[(size_t)j].data(), blocks[(size_t)j].size());
                const auto &r = recs[(size_t)j];
                if (r.empty()) {
                    fx.be(0, 8);
                    continue;
                }
                // record 0: (number of records, offset of the mask's first record << 1), then one per anchor partition present
                fx.be((uint64_t)r.size() + 1, 8);
                fx.be((uint64_t)r.size() + 1, 8);
                fx.be((uint64_t)(at + 8) << 1, 8);
                for (auto &e : r) {
                    fx.be(e.first, 8);
                    fx.be((((e.second >> 1) + (uint64_t)at) << 1) | (e.second & 1), 8);
                }
            }
            if (fd.bad || fx.bad) throw HipError("lm_index_save: write failed (seeds)");
        }
        return LM_OK;
    } catch (const std::exception &e) {
        ix->err = e.what();
        return LM_ERR_HIP;
    }
}
