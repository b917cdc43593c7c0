// lm_kernels.hip — gfx950 kernels of the `lexicmap search` hot path.
//
// Work decomposition:
//   k_extract_kmers     one lane per query position          -> 2-bit k-mers of both strands (+ filtered copy)
//   k_build_cmp_tab/bits per query                           -> bucket table + 8-base prefix bitmap of the filtered k-mers
//   k_mask              one lane per (query, mask)           -> LexicHash capture over the sorted k-mer array
//   k_lookup_prep/count/emit one lane per (query, mask, dir) -> prefix/suffix range query in the HBM seed arrays,
//                                                               in seed-list order, sampled top array first
//   k_chain1            one lane per (query, genome)         -> ClearSubstrPairs + Chainer.Chain
//   k_make_tasks        one lane per (query, genome)         -> chain windows
//   k_pa_filter         one workgroup per 64 chain windows   -> SeqComparator.Compare: positions that can match (LDS maps)
//   k_pa_search         one lane per candidate position      -> tree.Search emulation + anchors
//   k_pa_chain_wave     one wavefront per chain              -> Clear + Trim + Chainer2
//   k_extract_windows   one workgroup per chain with results -> 2-bit genome -> ASCII window (rc applied)
//   k_extend_count/k_extend/k_extend_fin one lane per HSP flank -> extendMatch (Chainer3 on the (q,t) grid)
//   k_wfa_lean<2>       persistent wavefronts, queue of HSPs -> WFA + backtrace + statistics + BLAST-style score
//   k_wfa_wave          one wavefront per HSP                -> the same with a global-memory ring (fallback)
// Radix sorts / scans / run-length encodes between kernels are rocPRIM device primitives (plumbing).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "lm_algos.h"
#include "lm_kernels.h"

namespace lm {

// Wavefront-private LDS: LDS traffic of a single wave is processed in issue order, so cross-lane LDS hand-offs
// only need the compiler not to reorder them. A workgroup barrier would also drain the outstanding global stores
// (s_waitcnt vmcnt(0)) of the backtrace store on every score step, which is what this avoids.
#define LDS_WAVE_SYNC()                                        \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)

// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int find_segment(const int64_t *off, int n, int64_t x) {
    // largest s with off[s] <= x  (off[0]=0, off[n]=total)
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (off[mid] <= x)
            lo = mid;
        else
            hi = mid - 1;
    }
    return lo;
}

__device__ __forceinline__ uint64_t encode_kmer(const uint8_t *s, int k) {
    uint64_t c = 0;
    for (int i = 0; i < k; i++) c = (c << 2) | lm_base2bit(s[i]);
    return c;
}

// posoff[q] = number of k-mer positions before query q; keys of query q live at [2*posoff[q], 2*posoff[q+1])
// K (<= 32) bases from 32 readable bytes: two 16-byte loads instead of K byte loads
__device__ __forceinline__ uint64_t encode_kmer32(const uint8_t *s, int k) {
    uint32_t w[8];
    __builtin_memcpy(w, s, 32);
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 32; i++) {
        const uint64_t b = lm_base2bit((uint8_t)(w[i >> 2] >> ((i & 3) << 3)));
        if (i < k) c = (c << 2) | b;
    }
    return c;
}
__global__ void k_extract_kmers(const uint8_t *__restrict__ qseq, const int64_t *__restrict__ qoff,
                                const int64_t *__restrict__ posoff, int nq, int K, uint64_t *__restrict__ keys_all,
                                uint32_t *__restrict__ vals_all, uint64_t *__restrict__ keys_cmp,
                                uint32_t *__restrict__ vals_cmp, int32_t *__restrict__ nvalid) {
    const int64_t total = posoff[nq], nbases = qoff[nq];
    const int64_t nround = (total + blockDim.x - 1) / blockDim.x * blockDim.x; // whole wavefronts stay together (ballot below)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nround; i += (int64_t)gridDim.x * blockDim.x) {
        const bool live = i < total;
        int q = 0;
        bool filtered = true;
        if (live) {
            q = find_segment(posoff, nq, i);
            const int pos = (int)(i - posoff[q]);
            const int64_t at = qoff[q] + pos;
            const uint64_t fwd = at + 32 <= nbases ? encode_kmer32(qseq + at, K) : encode_kmer(qseq + at, K);
            const uint64_t rc = lm_revcomp(fwd, K);
            keys_all[2 * i] = fwd;
            keys_all[2 * i + 1] = rc;
            vals_all[2 * i] = (uint32_t)pos << 1;
            vals_all[2 * i + 1] = ((uint32_t)pos << 1) | 1u;
            // SeqComparator.Index filter (lib-seq_compare.go:143): both strands of a position are dropped together
            filtered = fwd == 0 || lm_low_complexity(fwd, K);
            keys_cmp[2 * i] = filtered ? (1ull << 63) : fwd;
            keys_cmp[2 * i + 1] = filtered ? (1ull << 63) : rc;
            vals_cmp[2 * i] = (uint32_t)pos << 1;
            vals_cmp[2 * i + 1] = ((uint32_t)pos << 1) | 1u;
        }
        // the 64 positions of a wavefront nearly always belong to one query: one atomic for all of them (one per position
        // was ~20 M atomics on ~1000 addresses per C3 part - most of the kernel's 23 ms)
        const int q0 = __builtin_amdgcn_readfirstlane(q);
        const bool mine = live && !filtered;
        const uint64_t same = __ballot(mine && q == q0);
        if ((threadIdx.x & 63) == 0 && same) atomicAdd(&nvalid[q0], 2 * (int)__popcll(same));
        if (mine && q != q0) atomicAdd(&nvalid[q], 2);
    }
}

// bucket table over the leading tab_bits[q] bits of each query's sorted (filtered) k-mer array (sized per query so that a
// bucket holds about half a k-mer: 2^13 buckets for a gene, 2^17 for a 27-kb read), at tab + tab_off[q]:
// tab[b] = first index (relative to the query's segment) whose key >> (2K-bits) >= b; tab[2^bits] = nvalid[q]
__global__ void k_build_cmp_tab(const uint64_t *__restrict__ keys_cmp, const int64_t *__restrict__ posoff,
                                const int32_t *__restrict__ nvalid, int nq, int K, const int64_t *__restrict__ tab_off,
                                const int32_t *__restrict__ tab_bits, uint32_t *__restrict__ tab) {
    const int64_t total = tab_off[nq];
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int q = find_segment(tab_off, nq, t);
        const int bits = tab_bits[q], nb = 1 << bits;
        const int b = (int)(t - tab_off[q]);
        const uint64_t *keys = keys_cmp + 2 * posoff[q];
        int n = nvalid[q];
        int lo = n;
        if (b < nb) {
            uint64_t target = (uint64_t)b << ((K << 1) - bits);
            lo = lm_lower_bound_u64(keys, 0, n, target);
        }
        tab[t] = (uint32_t)lo;
    }
}
// Prefix filters of a query for the pseudo-alignment (layout and logic: lm_pa_filter_set / lm_pa_candidate2, lm_algos.h):
// hashed bitmaps of 2^log bits over the 11-base prefixes (the smallest prefix length SeqComparator.Compare ever asks for,
// lib-seq_compare.go:339-348) and the 9-base prefixes of its (filtered) k-mers, sized per query with ~16 bits per k-mer
// (a 1.5-kb gene: 64 Kbit, a 50-kb read: 2 Mbit), plus the two maps k_pa_filter keeps in LDS: a two-hash Bloom filter
// of the 11-base prefixes (<= 2^19 bits) and the exact 2^18-bit map of the 9-base prefixes.
__global__ __launch_bounds__(256) void k_build_cmp_bits(const uint64_t *__restrict__ keys_cmp,
                                                         const int64_t *__restrict__ posoff,
                                                         const int32_t *__restrict__ nvalid, int nq, int K,
                                                         const int64_t *__restrict__ bits_off,
                                                         const int32_t *__restrict__ bits_log, uint32_t *__restrict__ bits) {
    // the bitmaps were zeroed by the host; one workgroup per query sets its bits
    for (int q = blockIdx.x; q < nq; q += gridDim.x) {
        const uint64_t *keys = keys_cmp + 2 * posoff[q];
        const int n = nvalid[q];
        const int log = bits_log[q];
        uint32_t *b = bits + bits_off[q];
        for (int j = threadIdx.x; j < n; j += blockDim.x)
            lm_pa_filter_set(keys[j], K, log, [&](uint64_t w, uint32_t m) { atomicOr(&b[w], m); });
    }
}

__global__ void k_fill_u32(uint32_t *p, int64_t n, uint32_t v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

// ------------------------------------------------------------------------------------------------------------
// lexichash masking (lib-index-search.go:1212-1238)
__global__ void k_mask(const uint64_t *__restrict__ keys_all, const int64_t *__restrict__ posoff, int nq, int M, int K,
                       const uint64_t *__restrict__ masks, uint64_t *__restrict__ out_kmers,
                       int64_t *__restrict__ out_lo, int64_t *__restrict__ out_hi, uint32_t *__restrict__ first_mask) {
    int64_t total = (int64_t)nq * M;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        int q = (int)(t / M), m = (int)(t % M);
        int64_t base = 2 * posoff[q];
        int n = (int)(2 * (posoff[q + 1] - posoff[q]));
        uint64_t kmer = 0;
        int lo = 0, hi = 0;
        if (n > 0) {
            kmer = lm_xor_argmin(keys_all + base, n, masks[m], &lo, &hi);
            if (kmer != 0 && lm_low_complexity(kmer, K)) kmer = 0;
            if (kmer != 0) atomicMin(&first_mask[base + lo], (uint32_t)m);
        }
        out_kmers[t] = kmer;
        out_lo[t] = base + lo;
        out_hi[t] = base + hi;
    }
}

// ------------------------------------------------------------------------------------------------------------
// seed lookup (kv-searcher2.go:105-549 semantics over the flat HBM arrays) + anchor assembly (:1398-1562)
__device__ __forceinline__ int argmin_mask(const uint64_t *masks, const int32_t *pfx_first, int K, int p, uint64_t kmer) {
    uint64_t pf = kmer >> ((K - p) << 1);
    int minj = -1;
    uint64_t minh = ~0ull;
    for (int j = pfx_first[pf]; j < pfx_first[pf + 1]; j++) {
        uint64_t h = masks[j] ^ kmer;
        if (h < minh) {
            minh = h;
            minj = j;
        }
    }
    return minj;
}

// ---- seed lookup ------------------------------------------------------------------------------------------------------
// Packed seed image (DevIndexView, lm_seedpack.hip): list md = (mask, direction) -> partition table over the a bases that
// follow the mask's p-base prefix (the reference's anchor partitions, kv-data.go:90-125, 413-434) -> sorted key_bits-wide
// k-mer remainders + packed values.  A lookup reads one table entry pair, binary-searches ONE partition (tens of seeds:
// a few hundred bytes) and scans the matches: kv-searcher2.go:105-323 semantics ("all k-mers in [kmer & ~m, kmer | m]
// whose reversed flag equals the direction"; the range never leaves a partition because MinPrefix >= p + a,
// lib-index-search.go:483-485).  Captured k-mers that do not start with the mask's prefix (short queries: most masks)
// can only match seeds of the flat outlier lists, which exist for tiny genomes only; when the list is empty the lookup
// is not even issued.
// Lookups are compacted and sorted by (list, partition) so that neighbouring threads read neighbouring table entries
// and partitions, and logical workgroups are laid out so that a contiguous eighth of the sorted lookups runs on one
// XCD (each XCD has its own L2), see lookup_index().
#define LM_LK_OUTLIER_BIT 31
__device__ __forceinline__ int local_genome(const DevIndexView &ix, uint64_t bg) {
    uint64_t batch = bg >> 17, gi = bg & 0x1ffff;
    if (batch >= (uint64_t)ix.nbatches) return -1;
    int64_t g = ix.batch_first[batch] + (int64_t)gi;
    if (ix.g2local) return g < ix.batch_first[ix.nbatches] ? ix.g2local[g] : -1; // chunk-aware shard table (loader)
    if (ix.shard_count > 1) {
        if ((int)(g % ix.shard_count) != ix.shard_rank) return -1;
        g /= ix.shard_count;
    }
    if (g >= ix.ngenomes) return -1;
    return (int)g;
}
__device__ __forceinline__ bool genome_kept(const DevIndexView &ix, int64_t g) {
    return g >= 0 && ((ix.g_keep[g >> 5] >> (g & 31)) & 1u) != 0;
}
__device__ __forceinline__ uint32_t lookup_sort_key(const DevIndexView &ix, uint32_t md, uint64_t x) {
    const int p = ix.mask_prefix;
    const uint64_t mp = ix.masks[md >> 1] >> ((ix.K - p) << 1);
    if ((x >> ((ix.K - p) << 1)) == mp) {
        const uint32_t part = (uint32_t)(x >> ix.key_bits) & (uint32_t)(ix.P1 - 2);
        return (md << (ix.part_bases << 1)) | part;
    }
    if (ix.out_off[md + 1] > ix.out_off[md]) return (1u << LM_LK_OUTLIER_BIT) | md;
    return 0xffffffffu; // nothing stored under this list can share min_prefix >= p bases with x
}
// one slot per (query, mask, direction): t = (query*M + mask)*2 + dir.  A workgroup classifies a tile of LK_TILE
// consecutive slots, reserves room for its issued lookups with ONE atomic and appends them (key, slot); the radix sort
// that follows puts them in (list, partition) order.
#define LK_PER_THREAD 8
#define LK_TILE (256 * LK_PER_THREAD)
__global__ __launch_bounds__(256) void k_lookup_prep(DevIndexView ix, const uint64_t *__restrict__ kmers,
                                                     const int64_t *__restrict__ klo,
                                                     const uint32_t *__restrict__ first_mask, int64_t nqm,
                                                     uint32_t *__restrict__ keys, uint32_t *__restrict__ slots,
                                                     unsigned long long *__restrict__ counter) {
    __shared__ uint32_t wave_cnt[4];
    __shared__ unsigned long long tile_base;
    const int64_t total = nqm * 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t tile = (int64_t)blockIdx.x * LK_TILE; tile < total; tile += (int64_t)gridDim.x * LK_TILE) {
        uint32_t key[LK_PER_THREAD];
        uint32_t mine = 0;
#pragma unroll
        for (int r = 0; r < LK_PER_THREAD; r++) {
            const int64_t t = tile + r * 256 + threadIdx.x; // coalesced over the tile
            uint32_t k = 0xffffffffu;
            if (t < total) {
                const int dir = (int)(t & 1);
                const int64_t qm = t >> 1;
                const uint64_t kmer = kmers[qm];
                const int m = (int)(qm % ix.M);
                if (kmer != 0) {
                    if (dir == 0) {
                        k = lookup_sort_key(ix, (uint32_t)m << 1, kmer);
                    } else if (first_mask[klo[qm]] == (uint32_t)m) { // de-duplicated reversed k-mer (:1288-1298)
                        const uint64_t rev = lm_reverse(kmer, ix.K);
                        const int a = argmin_mask(ix.masks, ix.pfx_first, ix.K, ix.mask_prefix, rev);
                        if (a >= 0) k = lookup_sort_key(ix, ((uint32_t)a << 1) | 1u, rev);
                    }
                }
            }
            key[r] = k;
            mine += k != 0xffffffffu;
        }
        // exclusive prefix of `mine` over the workgroup: wave scan + 4 wave totals
        uint32_t incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        if (lane == 63) wave_cnt[wave] = incl;
        __syncthreads();
        uint32_t before = 0, all = 0;
        for (int w2 = 0; w2 < 4; w2++) {
            if (w2 < wave) before += wave_cnt[w2];
            all += wave_cnt[w2];
        }
        if (threadIdx.x == 0 && all) tile_base = atomicAdd(counter, (unsigned long long)all);
        __syncthreads();
        if (mine) {
            unsigned long long o = tile_base + before + (incl - mine);
#pragma unroll
            for (int r = 0; r < LK_PER_THREAD; r++)
                if (key[r] != 0xffffffffu) {
                    keys[o] = key[r];
                    slots[o] = (uint32_t)(tile + r * 256 + threadIdx.x);
                    o++;
                }
        }
        __syncthreads();
    }
}
// position in sorted order handled by this thread: logical workgroup x*per + y runs as hardware workgroup y*8 + x, i.e.
// XCD x owns one contiguous eighth of the sorted lookups
__device__ __forceinline__ int64_t lookup_index(int64_t total) {
    const int64_t per = (int64_t)(gridDim.x >> 3);
    const int64_t lb = (int64_t)(blockIdx.x & 7) * per + (int64_t)(blockIdx.x >> 3);
    const int64_t j = lb * blockDim.x + threadIdx.x;
    return j < total ? j : -1;
}
__device__ __forceinline__ void lookup_range(uint64_t key, int K, int min_prefix, uint64_t *left, uint64_t *right) {
    if (min_prefix < K) {
        const uint64_t low = (1ull << ((K - min_prefix) << 1)) - 1;
        *left = key & ~low;
        *right = key | low;
    } else {
        *left = *right = key;
    }
}
__global__ __launch_bounds__(256) void k_lookup_count(DevIndexView ix, const uint64_t *__restrict__ kmers,
                                                      const int64_t *__restrict__ klo, const int64_t *__restrict__ khi,
                                                      const uint32_t *__restrict__ skeys, const uint32_t *__restrict__ sslots,
                                                      int64_t nlk, int min_prefix, uint32_t *__restrict__ counts,
                                                      int64_t *__restrict__ starts, int32_t *__restrict__ nscan,
                                                      unsigned long long *__restrict__ stat_values,
                                                      uint64_t *__restrict__ lkey, uint64_t *__restrict__ lrec) {
    const int64_t j = lookup_index(nlk);
    if (j < 0) return;
    const uint32_t sk = skeys[j];
    const int64_t t = (int64_t)sslots[j];
    const int dir = (int)(t & 1);
    const int64_t qm = t >> 1;
    uint64_t key = kmers[qm];
    if (dir) key = lm_reverse(key, ix.K);
    uint64_t left, right;
    lookup_range(key, ix.K, min_prefix, &left, &right);
    int64_t st;
    int32_t nv = 0, nkept = 0;
    if (!(sk >> LM_LK_OUTLIER_BIT)) {
        const int pb = ix.part_bases << 1;
        const uint32_t md = sk >> pb, part = sk & ((1u << pb) - 1);
        const uint32_t *row = ix.part_tab + (int64_t)md * ix.P1 + part;
        const int64_t base = ix.md_off[md];
        const uint64_t km = (1ull << ix.key_bits) - 1;
        nv = lm_partition_range(ix.pk_keys, ix.key_bits, base + row[0], base + row[1], left & km, right & km, &st);
        if (ix.g_keep && nv) { // genome whitelist: count the kept seeds only (the emit pass skips the others)
            int32_t kept = 0;
            for (int32_t x = 0; x < nv; x++)
                kept += genome_kept(ix, (int64_t)lm_packed_val_genome(lm_bits_get(ix.pk_vals, st + x, ix.gid_bits + ix.pos_bits + 1), ix.pos_bits));
            nkept = kept;
        } else {
            nkept = nv;
        }
    } else {
        const uint32_t md = sk & 0x7fffffffu;
        int64_t lo = ix.out_off[md], hi = ix.out_off[md + 1];
        const int64_t e = hi;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (ix.out_kmers[mid] < left)
                lo = mid + 1;
            else
                hi = mid;
        }
        st = lo;
        while (lo < e && ix.out_kmers[lo] <= right) lo++;
        nv = (int32_t)(lo - st);
        nkept = nv;
        if (ix.g_keep && nv) {
            nkept = 0;
            for (int32_t x = 0; x < nv; x++) nkept += genome_kept(ix, local_genome(ix, ix.out_vals[st + x] >> 30));
        }
    }
    const int64_t l0 = klo[qm], l1 = khi[qm];
    counts[j] = (uint32_t)nkept * (uint32_t)(l1 - l0);
    starts[j] = st;
    nscan[j] = nv;
    // what k_lookup_emit_flat needs of this lookup, in sorted order (coalesced there): the looked-up k-mer and the range of
    // the query's locations of it - gathered by (query, mask) once, here, instead of once more per kernel
    lkey[j] = key;
    lrec[j] = ((uint64_t)(uint32_t)l0 << 32) | (uint64_t)(uint32_t)(l1 - l0);
    if (nkept) atomicAdd(stat_values, (unsigned long long)nkept);
}

__global__ __launch_bounds__(256) void k_lookup_emit(DevIndexView ix, const uint64_t *__restrict__ kmers,
                                                     const int64_t *__restrict__ klo, const int64_t *__restrict__ khi,
                                                     const uint32_t *__restrict__ vals_all,
                                                     const uint32_t *__restrict__ skeys, const uint32_t *__restrict__ sslots,
                                                     int64_t nlk, const uint32_t *__restrict__ counts,
                                                     const int64_t *__restrict__ offs, const int64_t *__restrict__ starts,
                                                     const int32_t *__restrict__ nscan, uint64_t *__restrict__ outA,
                                                     uint64_t *__restrict__ outB) {
    const int64_t j = lookup_index(nlk);
    if (j < 0 || counts[j] == 0) return;
    const uint32_t sk = skeys[j];
    const int64_t t = (int64_t)sslots[j];
    const int dir = (int)(t & 1);
    const int64_t qm = t >> 1;
    const uint64_t q = (uint64_t)(qm / ix.M);
    uint64_t key = kmers[qm];
    if (dir) key = lm_reverse(key, ix.K);
    int64_t o = offs[j];
    const int64_t b = starts[j];
    const bool outlier = (sk >> LM_LK_OUTLIER_BIT) != 0;
    const uint64_t km = (1ull << ix.key_bits) - 1;
    const int fixed = ix.K - (ix.key_bits >> 1); // p + a bases shared by construction
    for (int32_t s = 0; s < nscan[j]; s++) {
        uint64_t v;
        int kprefix;
        if (!outlier) {
            const uint64_t sr = lm_bits_get(ix.pk_keys, b + s, ix.key_bits);
            const uint64_t d = sr ^ (key & km);
            kprefix = fixed + (d ? ((lm_clz64(d) - (64 - ix.key_bits)) >> 1) : (ix.key_bits >> 1));
            const uint64_t pv = lm_bits_get(ix.pk_vals, b + s, ix.gid_bits + ix.pos_bits + 1);
            if (ix.g_keep && !genome_kept(ix, (int64_t)lm_packed_val_genome(pv, ix.pos_bits))) continue;
            v = lm_unpack_seed_val(pv, ix.g_bg[lm_packed_val_genome(pv, ix.pos_bits)], ix.pos_bits, dir);
        } else {
            v = ix.out_vals[b + s];
            if (ix.g_keep && !genome_kept(ix, local_genome(ix, v >> 30))) continue;
            kprefix = lm_lcp(key, ix.out_kmers[b + s], ix.K);
        }
        const uint64_t A = (q << 34) | (v >> 30);
        for (int64_t li = klo[qm]; li < khi[qm]; li++) {
            const uint32_t loc = vals_all[li];
            int bq, bt;
            bool rct;
            lm_anchor_coords(v, (int)(loc >> 1), (loc & 1u) != 0, kprefix, ix.K, &bq, &bt, &rct);
            outA[o] = A;
            outB[o] = lm_pack_anchor(bq, kprefix, bt, (loc & 1u) != 0, rct);
            o++;
        }
    }
}

// k_lookup_emit_flat: the same anchors with the lanes over the OUTPUT instead of the lookups.  The anchors of the 64 lookups of
// a wavefront are one contiguous range of the output (offs = exclusive scan in lookup order), so lane l takes anchors l, l + 64,
// ...: it finds the lookup an anchor belongs to by a binary search over the wavefront's 64 offsets (shuffles), takes that
// lookup's record from the lane that holds it, reads ONE seed and writes ONE anchor.  Stores are full lines, the packed key /
// value streams of a partition are read front to back by neighbouring lanes, a lookup that returns thousands of seeds is
// spread over the wavefront (k_lookup_emit: one lane walking them, 8-byte stores 16 bytes apart per lane: 16 GB written for
// 3.5 GB of anchors, 50 GB read - it re-gathered k-mer and location range by (query, mask) and touched every seed line once per
// lane).  Not for searches under a genome whitelist (the kept seeds of a lookup are then not a prefix of its range).
__global__ __launch_bounds__(256) void k_lookup_emit_flat(DevIndexView ix, const uint32_t *__restrict__ vals_all,
                                                          const uint32_t *__restrict__ skeys, const uint32_t *__restrict__ sslots,
                                                          int64_t nlk, const uint32_t *__restrict__ counts,
                                                          const int64_t *__restrict__ offs, const int64_t *__restrict__ starts,
                                                          const uint64_t *__restrict__ lkey, const uint64_t *__restrict__ lrec,
                                                          uint64_t *__restrict__ outA, uint64_t *__restrict__ outB) {
    const int64_t j = lookup_index(nlk);
    uint32_t cnt = 0, sk = 0, slot = 0;
    int64_t off = offs[nlk], st = 0; // (lanes behind the last lookup: the end of the output, no anchors)
    uint64_t key = 0, rec = 0;
    if (j >= 0) {
        cnt = counts[j];
        off = offs[j];
        if (cnt) {
            sk = skeys[j];
            slot = sslots[j];
            st = starts[j];
            key = lkey[j];
            rec = lrec[j];
        }
    }
    const int64_t wave_base = __shfl(off, 0, 64);
    const int64_t total = __shfl(off + (int64_t)cnt, 63, 64) - wave_base;
    const uint64_t km = (1ull << ix.key_bits) - 1;
    const int fixed = ix.K - (ix.key_bits >> 1); // p + a bases shared by construction
    auto one = [&](uint32_t sk_, uint32_t slot_, int64_t st_, uint64_t key_, uint64_t rec_, int64_t r, uint64_t *A, uint64_t *B) {
        const uint32_t nloc = (uint32_t)rec_;
        const int64_t s_i = nloc == 1 ? r : r / (int64_t)nloc;
        const uint32_t li = (uint32_t)(rec_ >> 32) + (uint32_t)(nloc == 1 ? 0 : r - s_i * (int64_t)nloc);
        const int dir = (int)(slot_ & 1u);
        const uint64_t q = (uint64_t)((slot_ >> 1) / (uint32_t)ix.M);
        uint64_t v;
        int kprefix;
        if (!(sk_ >> LM_LK_OUTLIER_BIT)) {
            const uint64_t sr = lm_bits_get(ix.pk_keys, st_ + s_i, ix.key_bits);
            const uint64_t d = sr ^ (key_ & km);
            kprefix = fixed + (d ? ((lm_clz64(d) - (64 - ix.key_bits)) >> 1) : (ix.key_bits >> 1));
            const uint64_t pv = lm_bits_get(ix.pk_vals, st_ + s_i, ix.gid_bits + ix.pos_bits + 1);
            v = lm_unpack_seed_val(pv, ix.g_bg[lm_packed_val_genome(pv, ix.pos_bits)], ix.pos_bits, dir);
        } else {
            v = ix.out_vals[st_ + s_i];
            kprefix = lm_lcp(key_, ix.out_kmers[st_ + s_i], ix.K);
        }
        const uint32_t loc = vals_all[li];
        int bq, bt;
        bool rct;
        lm_anchor_coords(v, (int)(loc >> 1), (loc & 1u) != 0, kprefix, ix.K, &bq, &bt, &rct);
        *A = (q << 34) | (v >> 30);
        *B = lm_pack_anchor(bq, kprefix, bt, (loc & 1u) != 0, rct);
    };
    if (total >= ((int64_t)1 << 31)) { // (never at the batch sizes the parts are cut to: the offsets below are 32-bit)
        for (int64_t r = 0; r < (int64_t)cnt; r++) one(sk, slot, st, key, rec, r, &outA[off + r], &outB[off + r]);
        return;
    }
    const uint32_t rel = (uint32_t)(off - wave_base);
    for (uint32_t base = 0; base < (uint32_t)total; base += 64) { // wave-uniform trip count: every lane takes part in the shuffles
        const uint32_t o = base + (uint32_t)(threadIdx.x & 63);
        const bool act = o < (uint32_t)total;
        int L = 0; // the last lane whose first anchor is <= o (lanes without anchors share their successor's offset)
#pragma unroll
        for (int step = 32; step > 0; step >>= 1) {
            const uint32_t rc = __shfl(rel, L + step, 64); // (L + step <= 63)
            if (rc <= o) L += step;
        }
        const uint32_t r = o - __shfl(rel, L, 64);
        const uint32_t sk_ = __shfl(sk, L, 64), slot_ = __shfl(slot, L, 64);
        const int64_t st_ = __shfl(st, L, 64);
        const uint64_t key_ = __shfl(key, L, 64), rec_ = __shfl(rec, L, 64);
        if (act) {
            uint64_t A, B;
            one(sk_, slot_, st_, key_, rec_, (int64_t)r, &A, &B);
            outA[wave_base + o] = A;
            outB[wave_base + o] = B;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// per (query,genome): ClearSubstrPairs + Chainer.Chain (lib-index-search.go:1702-1775)
// Small pairs (nearly all: the random 17-base matches of unrelated genomes give one to three anchors): one lane per pair.
// Pairs above LM_CHAIN1_WAVE_MIN anchors (a read against the members of its own family: hundreds of anchors; a 200-kb
// plasmid query: thousands) are left to k_chain1_wave - one lane walking n x window candidates alone kept a whole launch
// waiting (27 ms per C3 launch, 55 ms of a 360-ms C4-shaped step).
#define LM_CHAIN1_WAVE_MIN 48
__global__ void k_chain1(const uint64_t *__restrict__ B, const int64_t *__restrict__ seg_off, int nseg, LmChainOpt opt,
                         int K, LmSub *__restrict__ subs, uint8_t *__restrict__ marks, uint64_t *__restrict__ msi,
                         uint64_t *__restrict__ s2i, int8_t *__restrict__ dirs, uint8_t *__restrict__ visited,
                         int32_t *__restrict__ chain_off_pool, int32_t *__restrict__ chain_idx_pool,
                         int32_t *__restrict__ seg_n, float *__restrict__ seg_score, int32_t *__restrict__ seg_nch,
                         int32_t *__restrict__ big_list, unsigned int *__restrict__ big_count) {
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < nseg; s += gridDim.x * blockDim.x) {
        int64_t o = seg_off[s];
        int n = (int)(seg_off[s + 1] - o);
        if (big_list && n > LM_CHAIN1_WAVE_MIN) {
            big_list[atomicAdd(big_count, 1u)] = s;
            continue;
        }
        LmSub *sb = subs + o;
        for (int i = 0; i < n; i++) sb[i] = lm_unpack_anchor(B[o + i]);
        if (n > 1) n = lm_clear_sorted(sb, n, K, marks + o);
        int nch = 0;
        float sc = lm_run_chain1(sb, n, opt, msi + o, s2i + o, dirs + o, visited + o, chain_off_pool + o + 4ll * s,
                             chain_idx_pool + 2 * o + 8ll * s, &nch);
        seg_n[s] = n;
        seg_score[s] = sc;
        seg_nch[s] = nch;
    }
}

// One wavefront per large pair: ClearSubstrPairs with a lane per anchor (an anchor's mark depends on the ORIGINAL list
// only) and an ordered ballot compaction; Chainer.Chain's DP with the lanes over the candidate predecessors j (the scan
// from high j to low j with a strict `>` keeps the largest j among equal best scores = the maximum of (score bits, j)); the
// score list sorted by an all-ascending bitonic network in global memory (end-padded with +inf, so any n); the backtrack is
// the serial walk of lm_run_chain1.  Same results as lm_clear_sorted + lm_run_chain1 (the CPU-checked statement).
__device__ __forceinline__ unsigned long long chain1_wave_max_u64(unsigned long long v) {
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long x = __shfl_xor(v, o, 64);
        v = x > v ? x : v;
    }
    return v;
}
#define CHAIN1_WAVE_SYNC()           \
    do {                             \
        __threadfence_block();       \
        __builtin_amdgcn_wave_barrier(); \
    } while (0)
__global__ __launch_bounds__(256) void k_chain1_wave(const uint64_t *__restrict__ B, const int64_t *__restrict__ seg_off,
                                                      LmChainOpt opt, int K, LmSub *__restrict__ subs,
                                                      uint8_t *__restrict__ marks, uint64_t *__restrict__ msi,
                                                      uint64_t *__restrict__ s2i, int8_t *__restrict__ dirs,
                                                      uint8_t *__restrict__ visited, int32_t *__restrict__ chain_off_pool,
                                                      int32_t *__restrict__ chain_idx_pool, int32_t *__restrict__ seg_n,
                                                      float *__restrict__ seg_score, int32_t *__restrict__ seg_nch,
                                                      const int32_t *__restrict__ big_list,
                                                      const unsigned int *__restrict__ big_count) {
    const int lane = threadIdx.x & 63;
    const uint64_t lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
    const int nbig = (int)*big_count;
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    for (int bi = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); bi < nbig; bi += nwaves) {
        const int s = big_list[bi];
        const int64_t o = seg_off[s];
        const int n0 = (int)(seg_off[s + 1] - o);
        LmSub *sb = subs + o;
        uint8_t *mk = marks + o;
        uint64_t *ms = msi + o, *s2 = s2i + o;
        int8_t *dr = dirs + o;
        uint8_t *vis = visited + o;
        for (int i = lane; i < n0; i += 64) sb[i] = lm_unpack_anchor(B[o + i]);
        CHAIN1_WAVE_SYNC();
        // ---- ClearSubstrPairs (lm_clear_sorted): marks from the original list, then ordered compaction
        for (int i = lane; i < n0; i += 64) {
            uint8_t m = 0;
            if (i >= 1) {
                const LmSub v = sb[i];
                const int32_t vqend = v.qbegin + v.len;
                int32_t upbound = vqend - K;
                if (upbound < 0) upbound = 0;
                const int32_t vtbegin = v.tbegin, vtend = v.tbegin + v.len;
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
                    if (vqend <= p.qbegin + p.len && vtbegin >= p.tbegin && vtend <= p.tbegin + p.len) {
                        m = 1;
                        break;
                    }
                }
            }
            mk[i] = m;
        }
        CHAIN1_WAVE_SYNC();
        int n = 0;
        for (int c0 = 0; c0 < n0; c0 += 64) {
            const int i = c0 + lane;
            const bool keep = i < n0 && mk[i] == 0;
            LmSub v;
            if (i < n0) v = sb[i];
            CHAIN1_WAVE_SYNC(); // every lane holds its anchor before slots <= i are overwritten
            const uint64_t km = __ballot(keep);
            if (keep) sb[n + __popcll(km & lt_mask)] = v;
            n += __popcll(km);
        }
        CHAIN1_WAVE_SYNC();
        // ---- Chainer.Chain (lm_run_chain1)
        int32_t *chain_off = chain_off_pool + o + 4ll * s, *chain_idx = chain_idx_pool + 2 * o + 8ll * s;
        int nchains = 0, nidx = 0;
        float result = 0;
        if (lane == 0) chain_off[0] = 0;
        if (n == 1) {
            const float w = lm_seed_weight((float)sb[0].len);
            if (w >= opt.min_score && lane == 0) {
                chain_idx[0] = 0;
                chain_off[1] = 1;
            }
            nchains = w >= opt.min_score ? 1 : 0;
            result = w;
        } else {
            if (lane == 0) {
                const float s0 = lm_seed_weight((float)sb[0].len);
                ms[0] = (uint64_t)lm_f32bits(s0) << 32;
                dr[0] = 0;
                s2[0] = (uint64_t)lm_f32bits(s0) << 32;
            }
            const int32_t max_dist_i = (int32_t)opt.max_distance;
            for (int i = 1; i < n; i++) {
                CHAIN1_WAVE_SYNC(); // msi / dirs of i - 1 are visible
                const LmSub a = sb[i];
                const int32_t aq = a.qbegin, alen = a.len;
                const float m0 = lm_seed_weight((float)alen);
                int64_t tlo = (int64_t)a.tbegin - max_dist_i;
                if (a.tbegin < max_dist_i) tlo = 0;
                const int64_t thi = (int64_t)a.tbegin + max_dist_i;
                unsigned long long best = 0;
                for (int j0 = i - 1; j0 >= 0; j0 -= 64) {
                    const int j = j0 - lane;
                    bool within = false;
                    LmSub b;
                    b.qbegin = b.tbegin = 0;
                    b.len = 0;
                    if (j >= 0) {
                        b = sb[j];
                        within = aq - b.qbegin <= max_dist_i;
                    }
                    if (__ballot(within) == 0ull) break; // sorted by QBegin: everything further down is out as well
                    if (!within) continue;
                    if ((int64_t)b.tbegin < tlo || (int64_t)b.tbegin > thi) continue;
                    if (a.qbegin == b.qbegin || a.tbegin == b.tbegin) continue;
                    int32_t dq = a.qbegin - b.qbegin;
                    if (dq < 0) dq = -dq;
                    int32_t dt;
                    if (a.tbegin >= b.tbegin)
                        dt = a.tbegin - b.tbegin;
                    else
                        dt = a.tbegin + (int32_t)a.len - b.tbegin - (int32_t)b.len;
                    if (dt < 0) dt = -dt;
                    int32_t gi = dq - dt;
                    if (gi < 0) gi = -gi;
                    if ((float)gi > opt.max_gap) continue;
                    int32_t length;
                    float w;
                    if (aq > b.qbegin + (int32_t)b.len) {
                        length = alen;
                        w = lm_seed_weight((float)length);
                    } else if (gi == 0) {
                        length = aq + alen - b.qbegin;
                        w = -lm_seed_weight((float)b.len) + lm_seed_weight((float)length);
                    } else {
                        length = aq + alen - (b.qbegin + (int32_t)b.len);
                        w = lm_seed_weight((float)length);
                    }
                    const int dir = a.tbegin >= b.tbegin ? 1 : -1;
                    const float gs = gi < opt.gap_lut_n ? opt.gap_lut[gi] : 0.0f;
                    const int8_t dj = dr[j];
                    float sc;
                    if (dj == 0 || dj == dir) {
                        const float t = lm_f32frombits((uint32_t)(ms[j] >> 32)) + w;
                        sc = t - gs;
                    } else {
                        const float t = lm_seed_weight((float)b.len) + w;
                        sc = t - gs;
                    }
                    if (sc >= opt.min_score && sc > m0) { // positive floats: the bit pattern orders like the value
                        const unsigned long long key = ((unsigned long long)lm_f32bits(sc) << 32) | ((unsigned long long)(uint32_t)j << 1) |
                                                       (dir > 0 ? 1ull : 0ull);
                        best = key > best ? key : best;
                    }
                }
                best = chain1_wave_max_u64(best);
                if (lane == 0) {
                    float m = m0;
                    int mj = i;
                    int8_t mdir = 0;
                    if (best != 0ull) {
                        m = lm_f32frombits((uint32_t)(best >> 32));
                        mj = (int)((uint32_t)best >> 1);
                        mdir = (best & 1ull) ? 1 : -1;
                    }
                    ms[i] = ((uint64_t)lm_f32bits(m) << 32) | (uint32_t)mj;
                    dr[i] = mdir;
                    s2[i] = ((uint64_t)lm_f32bits(m) << 32) | (uint32_t)i;
                }
            }
            CHAIN1_WAVE_SYNC();
            // ---- the score list ascending: all-ascending bitonic network, elements beyond n count as +inf and never move
            for (int i = lane; i < n; i += 64) vis[i] = 0;
            int N = 1;
            while (N < n) N <<= 1;
            for (int k = 2; k <= N; k <<= 1) {
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int x = lane; x < (N >> 1); x += 64) {
                        const int i = ((x & ~(j - 1)) << 1) | (x & (j - 1)); // bit j clear
                        const int l = (j == (k >> 1)) ? (i ^ (k - 1)) : (i | j); // first step of a merge: mirrored partner
                        const int lo_i = i < l ? i : l, hi_i = i < l ? l : i;
                        if (hi_i < n) {
                            const uint64_t u = s2[lo_i], v = s2[hi_i];
                            if (u > v) {
                                s2[lo_i] = v;
                                s2[hi_i] = u;
                            }
                        }
                    }
                    CHAIN1_WAVE_SYNC();
                }
            }
            // ---- backtrack (serial walk, lm_run_chain1 :400-451); lane 0 walks, the others wait
            if (lane == 0) {
                int imax = n - 1;
                float max_score = 0;
                bool first = true;
                int nchecked = 0;
                while (true) {
                    nchecked++;
                    if (opt.top_chains > 0 && nchecked > opt.top_chains) break;
                    float M = 0;
                    uint32_t Mi = 0;
                    while (imax >= 0) {
                        M = lm_f32frombits((uint32_t)(s2[imax] >> 32));
                        Mi = (uint32_t)s2[imax];
                        if (!vis[Mi]) {
                            imax--;
                            break;
                        }
                        imax--;
                    }
                    if (M < opt.min_score) break;
                    const int pstart = nidx;
                    int i = (int)Mi;
                    if (first) {
                        max_score = M;
                        first = false;
                    }
                    while (true) {
                        const int j = (int)(ms[i] & 4294967295ull);
                        const bool change = (i != j && dr[j] != 0 && dr[i] != dr[j]);
                        if (vis[j] && !change) {
                            nidx = pstart;
                            vis[i] = 1;
                            break;
                        }
                        chain_idx[nidx++] = i;
                        vis[i] = 1;
                        if (i == j || change) {
                            if (change) chain_idx[nidx++] = j;
                            for (int x = pstart, y = nidx - 1; x < y; x++, y--) {
                                const int32_t t = chain_idx[x];
                                chain_idx[x] = chain_idx[y];
                                chain_idx[y] = t;
                            }
                            chain_off[++nchains] = nidx;
                            break;
                        } else {
                            i = j;
                        }
                    }
                }
                result = max_score;
            }
        }
        if (lane == 0) {
            seg_n[s] = n;
            seg_score[s] = result;
            seg_nch[s] = nchains;
        }
        CHAIN1_WAVE_SYNC();
    }
}

// ------------------------------------------------------------------------------------------------------------
// chain windows (lib-index-search.go:1966-2051)
__global__ void k_task_count(const float *__restrict__ seg_score, const int32_t *__restrict__ seg_nch,
                             const uint8_t *__restrict__ keep, int nseg, float min_score, int32_t *__restrict__ ntask) {
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < nseg; s += gridDim.x * blockDim.x)
        ntask[s] = (seg_score[s] >= min_score && (!keep || keep[s])) ? seg_nch[s] : 0;
}

__global__ void k_make_tasks(DevIndexView ix, const uint64_t *__restrict__ segA, const int64_t *__restrict__ seg_off,
                             int nseg, const LmSub *__restrict__ subs, const int32_t *__restrict__ chain_off_pool,
                             const int32_t *__restrict__ chain_idx_pool, const int32_t *__restrict__ ntask,
                             const int64_t *__restrict__ task_off, const int64_t *__restrict__ qoff, int ext_len,
                             int32_t *__restrict__ order_scratch, Task *__restrict__ tasks) {
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < nseg; s += gridDim.x * blockDim.x) {
        int nch = ntask[s];
        if (nch == 0) continue;
        int64_t o = seg_off[s];
        const LmSub *sb = subs + o;
        const int32_t *coff = chain_off_pool + o + 4ll * s;
        const int32_t *cidx = chain_idx_pool + 2 * o + 8ll * s;
        int32_t *order = order_scratch + o + 4ll * s;
        // stable sort of chains by TBegin of the first anchor (:1967-1974)
        for (int i = 0; i < nch; i++) {
            int32_t tx = sb[cidx[coff[i]]].tbegin;
            int j = i - 1;
            while (j >= 0 && sb[cidx[coff[order[j]]]].tbegin > tx) {
                order[j + 1] = order[j];
                j--;
            }
            order[j + 1] = i;
        }
        uint64_t A = segA[s];
        uint32_t q = (uint32_t)(A >> 34);
        uint64_t bg = A & ((1ull << 34) - 1);
        int g = local_genome(ix, bg);
        int qlen = (int)(qoff[q + 1] - qoff[q]);
        int glen = g >= 0 ? ix.g_len[g] : 0;
        for (int ci = 0; ci < nch; ci++) {
            int c = order[ci];
            const int32_t *chain = cidx + coff[c];
            int nseeds = coff[c + 1] - coff[c];
            LmSub first = sb[chain[0]], last = sb[chain[nseeds - 1]];
            int qb = first.qbegin, tb = first.tbegin;
            int qe = last.qbegin + last.len - 1, te = last.tbegin + last.len - 1;
            bool rc = nseeds == 1 ? (last.qrc != last.trc) : (tb > last.tbegin);
            int tBegin, tEnd;
            if (rc) {
                tBegin = last.tbegin - ext_len;
                if (tBegin < 0) tBegin = 0;
                tEnd = tb + last.len - 1 + ext_len;
            } else {
                tBegin = tb - ext_len;
                if (tBegin < 0) tBegin = 0;
                tEnd = te + ext_len;
            }
            int qBegin = qb - (qb < ext_len ? qb : ext_len);
            int qEnd = qe + (qlen - qe - 1 < ext_len ? qlen - qe - 1 : ext_len);
            // SubSeq3 clamping (genome.go:944-952) and the tEnd fix-up (:2045-2047)
            int st = tBegin, en = tEnd;
            if (en >= glen - 1) en = glen - 1;
            if (en < st) en = st;
            int wlen = g >= 0 && glen > 0 && st < glen ? en - st + 1 : 0;
            if (wlen < tEnd - tBegin + 1) tEnd -= tEnd - tBegin + 1 - wlen;
            Task t;
            t.seg = (uint32_t)s;
            t.bg = bg;
            t.q = q;
            t.g = g;
            t.rc = rc ? 1 : 0;
            t.tBegin = tBegin;
            t.tEnd = tEnd;
            t.qBegin = qBegin;
            t.qEnd = qEnd;
            t.nseeds = nseeds;
            t.wlen = wlen;
            t.woff = 0;
            tasks[task_off[s] + ci] = t;
        }
    }
}

__global__ void k_sum_i32(const int32_t *__restrict__ v, int64_t n, unsigned long long *__restrict__ out) {
    unsigned long long acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) acc += (unsigned long long)v[i];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}
__global__ void k_task_wlen(const Task *__restrict__ tasks, int64_t ntasks, int32_t *__restrict__ wlen) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ntasks; i += (int64_t)gridDim.x * blockDim.x)
        wlen[i] = tasks[i].wlen;
}
__global__ void k_task_set_woff(Task *__restrict__ tasks, int64_t ntasks, const int64_t *__restrict__ woff) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ntasks; i += (int64_t)gridDim.x * blockDim.x)
        tasks[i].woff = woff[i];
}

// genome.SubSeq3 (genome.go:931-1143) + RC (:2943) — one workgroup per chain window
// `only` (may be null): per task, > 0 when its window is needed (tasks with pseudo-alignment results: extendMatch and WFA
// read the ASCII window, the pseudo-alignment itself takes its k-mers from the packed genome)
__global__ void k_extract_windows(DevIndexView ix, const Task *__restrict__ tasks, int64_t ntasks,
                                  const int32_t *__restrict__ only, uint8_t *__restrict__ wbuf) {
    for (int64_t ti = blockIdx.x; ti < ntasks; ti += gridDim.x) {
        if (only && only[ti] <= 0) continue;
        const Task t = tasks[ti];
        if (t.wlen <= 0 || t.g < 0) continue;
        const uint8_t *gb = ix.gbits + ix.g_off[t.g];
        uint8_t *w = wbuf + t.woff;
        for (int i = threadIdx.x; i < t.wlen; i += blockDim.x) {
            int pos = t.rc ? (t.tBegin + t.wlen - 1 - i) : (t.tBegin + i);
            uint32_t code = (gb[pos >> 2] >> ((3 - (pos & 3)) << 1)) & 3u;
            if (t.rc) code = 3u - code;
            w[i] = (uint8_t)("ACGT"[code]);
        }
    }
}

// the same for a list of tasks, each window written at its own offset of a compact buffer (the windows of the tasks that
// produced pseudo-alignment chains, gathered over several chunks for one extendMatch / WFA round)
__global__ void k_extract_windows_at(DevIndexView ix, const Task *__restrict__ tasks, const int32_t *__restrict__ idx,
                                     const int64_t *__restrict__ dest, int64_t n, uint8_t *__restrict__ wbuf) {
    for (int64_t li = blockIdx.x; li < n; li += gridDim.x) {
        const Task t = tasks[idx[li]];
        if (t.wlen <= 0 || t.g < 0) continue;
        const uint8_t *gb = ix.gbits + ix.g_off[t.g];
        uint8_t *w = wbuf + dest[li];
        for (int i = threadIdx.x; i < t.wlen; i += blockDim.x) {
            int pos = t.rc ? (t.tBegin + t.wlen - 1 - i) : (t.tBegin + i);
            uint32_t code = (gb[pos >> 2] >> ((3 - (pos & 3)) << 1)) & 3u;
            if (t.rc) code = 3u - code;
            w[i] = (uint8_t)("ACGT"[code]);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// SeqComparator.Compare anchor generation (lib-seq_compare.go:335-445)
__device__ __forceinline__ int pa_min_prefix(int base, int wlen) {
    if (wlen >= 1000000) return base + 8;
    if (wlen >= 250000) return base + 6;
    if (wlen >= 50000) return base + 4;
    if (wlen >= 10000) return base + 2;
    return base;
}

// k-mer starting at base `pos` of a 2-bit packed genome (4 bases per byte, first base in the top bits): two aligned
// 64-bit loads instead of K byte loads from the ASCII window. The genome store is padded so the second word exists.
__device__ __forceinline__ uint64_t kmer_from_bits(const uint8_t *__restrict__ gbits, int64_t goff, int pos, int K) {
    const int64_t byte = goff + (pos >> 2);
    const uint64_t *p = (const uint64_t *)(gbits + (byte & ~7ll));
    const uint64_t H = __builtin_bswap64(p[0]), L = __builtin_bswap64(p[1]);
    const int o = (int)(byte & 7) * 8 + (pos & 3) * 2; // 0..62
    const uint64_t v = o ? ((H << o) | (L >> (64 - o))) : H;
    return v >> (64 - 2 * K);
}

// window k-mer at position i (and its reverse complement), from the packed genome when the task has one
__device__ __forceinline__ void pa_kmer(const Task &t, const uint8_t *__restrict__ w, const uint8_t *__restrict__ gbits,
                                        int64_t goff, int i, int K, uint64_t *kmer, uint64_t *rc) {
    if (gbits) {
        if (t.rc) {
            uint64_t g = kmer_from_bits(gbits, goff, t.tBegin + t.wlen - K - i, K);
            *rc = g;
            *kmer = lm_revcomp(g, K);
        } else {
            uint64_t g = kmer_from_bits(gbits, goff, t.tBegin + i, K);
            *kmer = g;
            *rc = lm_revcomp(g, K);
        }
    } else {
        *kmer = encode_kmer(w + i, K);
        *rc = lm_revcomp(*kmer, K);
    }
}

// The first P bases (P <= 16) of the window k-mer at position i and of its reverse complement, straight from the packed
// genome: all the per-position test of k_pa_filter needs (11-base filter prefix + the bases [7, P) of the
// partial-prefix rule). About half the arithmetic of building both full k-mers.
__device__ __forceinline__ uint32_t revcomp_small(uint32_t x, int P) { // P bases in the low 2P bits
    uint32_t y = __builtin_bitreverse32(~x);
    y = ((y >> 1) & 0x55555555u) | ((y & 0x55555555u) << 1);
    return y >> (32 - 2 * P);
}
__device__ __forceinline__ void pa_prefixes(const Task &t, const uint8_t *__restrict__ gbits, int64_t goff, int i, int K,
                                            int P, uint32_t *fwd, uint32_t *rc) {
    const int pos = t.rc ? t.tBegin + t.wlen - K - i : t.tBegin + i;
    const int64_t byte = goff + (pos >> 2);
    const uint64_t *p = (const uint64_t *)(gbits + (byte & ~7ll));
    const uint64_t H = __builtin_bswap64(p[0]), L = __builtin_bswap64(p[1]);
    const int o = (int)(byte & 7) * 8 + (pos & 3) * 2; // 0..62
    const uint64_t v = o ? ((H << o) | (L >> (64 - o))) : H; // 32 bases from `pos`, left aligned
    const uint32_t first = (uint32_t)(v >> (64 - 2 * P));
    const uint32_t last = (uint32_t)(v >> (64 - 2 * K)) & ((1u << (2 * P)) - 1u);
    const uint32_t rl = revcomp_small(last, P);
    *fwd = t.rc ? rl : first;
    *rc = t.rc ? first : rl;
}

// ---- pseudo-alignment anchors: filter + search ---------------------------------------------------------------------------
// Two kernels replace a count / scan / emit triple over the window positions.
//
// k_pa_filter: a workgroup of 1024 threads takes PA_GROUP consecutive chain windows (tasks are in (query, genome) order, so
// nearly always windows of ONE query) and keeps that query's Bloom filter of 11-base k-mer prefixes and its exact 9-base
// prefix map in LDS (96 KB, lm_pa_candidate2).  Per window position a lane extracts the first p bases of the k-mer and of
// its reverse complement from the 2-bit genome and asks the LDS maps: at 10^5 genomes nine windows in ten belong to
// unrelated genomes (random 17-base seed matches) and ~97 % of their positions end there, without a global load beyond the
// (coalesced) genome words.  What the Bloom filter lets through asks the query's 11-base bitmap in global memory when that
// is the more selective one (reads above ~16 kb).  The survivors - some query k-mer shares 11 bases, or the partial-prefix
// rule of tree.Search could fire - are appended to a candidate list (task, position, strand).  Wavefronts work
// independently (no workgroup barrier except when the query changes): each stages its candidates in its own LDS strip and
// appends them to the global list with one atomic per ~200.
//
// k_pa_search: one lane per candidate, no LDS, full occupancy (the exact search is a chain of ~10 dependent loads):
// lm_tree_search_range_tab + the enumeration of the matches; a wavefront reserves room for all its anchors with one atomic.
// Anchor order is irrelevant because the list is sorted by (task, B) afterwards.  Both counters keep counting past their
// capacity, so the host can re-run with larger buffers.
#define PA_THREADS 1024
#define PA_WAVES (PA_THREADS / 64)
static_assert(PA_WAVES == LM_PA_RANGE_SEGS, "one candidate segment per wavefront of a filter workgroup");
#define PA_GROUP LM_PA_GROUP /* chain windows per workgroup pass */
#define PA_STAGE 192 /* candidates a wavefront stages in LDS */
#define PA_SLICE 1920 /* window positions a wavefront takes at a time: their 2-bit genome words are one 8-byte load per lane */
#define PA_PEND 128 /* positions a wavefront sets aside for the global bitmap: examined whenever 64 have gathered */
#define PA_LDS_BYTES ((1 << (LM_PA_BLOOM_LOG_MAX - 3)) + (1 << (LM_PA_MAP9_LOG - 3)) + PA_WAVES * PA_STAGE * 8 + PA_WAVES * PA_PEND * 12)
// ROLL (K == 31): a lane takes ceil(np / 64) CONSECUTIVE positions of a slice instead of every 64th, so the 32-bit windows
// that hold the first p bases of a k-mer and of its reverse complement are funnel shifts by IMMEDIATES over the lane's own
// 64-base string (and its reverse complement, built once per slice) - no cross-lane traffic, no 64-bit shifts, no per-position
// reverse complement: ~50 vector instructions per position pair where the strided form (ROLL = false, kept for K != 31) has ~100.
#define PA_SLICE_ROLL 2048 /* 64 lanes x 32 positions */
// reverse complement of the 16 bases of a dword (first base in the top bits)
__device__ __forceinline__ uint32_t pa_rc16(uint32_t x) {
    const uint32_t y = __builtin_bitreverse32(~x);
    return ((y >> 1) & 0x55555555u) | ((y & 0x55555555u) << 1);
}
// lm_pa_candidate2 for the lanes `q` of a wavefront whose key f (first p bases) FAILED its own Bloom test, has bases [9, p) all
// A and belongs to a query without a global 11-base bitmap (log <= blog); the other lanes keep `c`.  One lane in eight is such
// a lane, so EVERY pass of a wavefront comes here, and the literal rule - three sibling prefixes, two hashes and two LDS reads
// each, per strand - was 3/4 of the kernel's instructions (217 per window base where the common path has ~50).  Staged instead:
// bases [7, p) all A - a candidate; otherwise the exact 9-base map must hit (a sibling shares 10 bases with the key, so its
// 9-base prefix is the key's: one LDS read turns away >= 88 % of the lanes); base 8 an A - that is the rule; base 8 not an A -
// the three sibling tests, entered only when some lane is left.
__device__ __forceinline__ bool pa_partial_rule(bool q, bool c, uint32_t f, int p, int blog, const uint32_t *bloom, const uint32_t *map9) {
    const uint32_t p9 = f >> ((p - 9) << 1);
    const bool all7 = (f & ((1u << ((p - 7) << 1)) - 1u)) == 0;
    const bool m9 = q && ((map9[p9 >> 5] >> (p9 & 31)) & 1u) != 0;
    const bool need = m9 && !all7 && (p9 & 3u) != 0;
    bool r = q ? (all7 || (m9 && (p9 & 3u) == 0)) : c;
    if (__ballot(need) != 0ull && need) {
        const uint32_t p11 = f >> ((p - LM_PFX_BASES) << 1);
        bool any = false;
#pragma unroll
        for (uint32_t sib = 1; sib < 4; sib++) {
            const uint32_t x = p11 | sib, sa = lm_pa_bloom_slot(x, 0, blog), sb = lm_pa_bloom_slot(x, 1, blog);
            any = any || (((bloom[sa >> 5] >> (sa & 31)) & (bloom[sb >> 5] >> (sb & 31))) & 1u) != 0;
        }
        r = any;
    }
    return r;
}
template <bool ROLL>
__global__ __launch_bounds__(PA_THREADS) void k_pa_filter(DevIndexView ix, const Task *__restrict__ tasks, int64_t ntasks,
                                                           const uint8_t *__restrict__ wbuf,
                                                           const int64_t *__restrict__ posoff,
                                                           const int32_t *__restrict__ nvalid,
                                                           const uint32_t *__restrict__ cmp_bits,
                                                           const int64_t *__restrict__ bits_off,
                                                           const int32_t *__restrict__ bits_log, int K, int min_prefix,
                                                           unsigned long long *__restrict__ seg_count, int nseg,
                                                           int64_t seg_cap, uint64_t *__restrict__ cand,
                                                           unsigned long long *__restrict__ group_counter, int seg_by_group) {
    // the candidate list is kept as `nseg` segments of `seg_cap` entries with a counter each: a single counter is a single
    // address in one L2 channel, and the ~10^6 appends of a launch (one per ~200 candidates) then queue up behind each other
    // for longer than all the rest of the kernel takes.  seg_by_group: nseg = R x PA_WAVES, the segment is chosen by the RANGE
    // of the task group (groups are in (query, genome) order and are handed out in order, so the PA_WAVES segments of a range
    // hold the candidates of one or two queries and k_pa_search can keep that query's tables in one XCD's L2) and by the
    // wavefront's number inside its workgroup (one counter per range was 160 wavefronts deep in atomics: the filter took
    // twice as long); otherwise by the wavefront alone (any query anywhere).
    extern __shared__ uint64_t pa_lds[];                               // PA_LDS_BYTES, dynamic (above 64 KB)
    uint64_t *s_stage = pa_lds;                                        // [PA_WAVES][PA_STAGE]
    uint32_t *s_bloom = (uint32_t *)(pa_lds + PA_WAVES * PA_STAGE);    // 64 KB
    uint32_t *s_map9 = s_bloom + (1 << (LM_PA_BLOOM_LOG_MAX - 5));     // 32 KB
    uint64_t *s_pend_rec = (uint64_t *)(s_map9 + (1 << (LM_PA_MAP9_LOG - 5))); // [PA_WAVES][PA_PEND]
    uint32_t *s_pend_pf = (uint32_t *)(s_pend_rec + PA_WAVES * PA_PEND);        // [PA_WAVES][PA_PEND]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint64_t *stg = s_stage + wave * PA_STAGE;
    int n_stg = 0; // wave-uniform
    const uint64_t lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
    int seg = (int)(((int64_t)blockIdx.x * PA_WAVES + wave) % nseg);
    uint64_t *seg_list = cand + (int64_t)seg * seg_cap;
    auto flush = [&]() { // this wavefront's staged candidates -> its segment of the global list
        if (n_stg == 0) return;
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(&seg_count[seg], (unsigned long long)n_stg);
        base = __shfl(base, 0, 64);
        for (int j = lane; j < n_stg; j += 64)
            if ((int64_t)(base + (unsigned long long)j) < seg_cap) seg_list[base + (unsigned long long)j] = stg[j];
        n_stg = 0;
    };
    auto push = [&](bool c, uint64_t rec) { // all lanes of the wavefront
        const uint64_t m = __ballot(c);
        if (m == 0) return;
        if (c) stg[n_stg + __popcll(m & lt_mask)] = rec;
        n_stg += __popcll(m);
        if (n_stg > PA_STAGE - 128) flush(); // (room for the 128 of the two-strand append) LDS accesses of one wavefront complete in program order
    };
    // pending list of the wavefront (long reads): first p bases (+ p) and the candidate record of the positions whose test
    // needs the global 11-base bitmap; a full 64 are examined at once.  log / blog / qbits belong to the run of one query:
    // the list is emptied before the run ends.
    uint32_t *pend_pf = s_pend_pf + wave * PA_PEND;
    uint64_t *pend_rec = s_pend_rec + wave * PA_PEND;
    int n_pend = 0; // wave-uniform
    const uint32_t *cur_qbits = nullptr;
    int cur_log = 0, cur_blog = 0;
    auto pend_round = [&](int cnt) { // the last `cnt` (<= 64) pending entries
        const int idx = n_pend - cnt + lane;
        bool c = false;
        uint64_t rec = 0;
        if (lane < cnt) {
            const uint32_t e = pend_pf[idx];
            rec = pend_rec[idx];
            c = lm_pa_candidate2(s_bloom, cur_blog, s_map9, cur_qbits, cur_log, e & 0x3fffffffu, LM_PFX_BASES + 2 * (int)(e >> 30));
        }
        n_pend -= cnt;
        push(c, rec);
    };
    auto pend = [&](bool c, uint32_t e, uint64_t rec) { // all lanes of the wavefront
        const uint64_t m = __ballot(c);
        if (m == 0) return;
        if (c) {
            const int slot = n_pend + __popcll(m & lt_mask);
            pend_pf[slot] = e;
            pend_rec[slot] = rec;
        }
        n_pend += __popcll(m);
        while (n_pend >= 64) pend_round(64);
    };
    // per group: the tasks' fields every wavefront needs (one round of dependent global loads for the whole group instead
    // of one per task and wavefront), then slices of PA_SLICE window positions handed out through an LDS counter, so the 16
    // wavefronts are in different windows at different stages and hide each other's genome-load latency
    __shared__ int32_t s_q[PA_GROUP], s_rc[PA_GROUP], s_tb[PA_GROUP], s_wlen[PA_GROUP], s_npos[PA_GROUP], s_log[PA_GROUP];
    __shared__ int32_t s_first[PA_GROUP + 1]; // first slice of every task of the current run of one query
    __shared__ int64_t s_goff[PA_GROUP], s_bits[PA_GROUP], s_woff[PA_GROUP];
    __shared__ int s_next;
    // persistent workgroups (one per CU: the 140 KB of LDS allow no second one) that take the groups from a global counter:
    // a workgroup per group left the CUs empty most of the time, between the end of one 16-wavefront workgroup and the
    // start of the next
    __shared__ unsigned long long s_grp;
    const int64_t ngroups = (ntasks + PA_GROUP - 1) / PA_GROUP;
    while (true) {
        __syncthreads(); // the previous group is finished (tables, s_grp)
        if (tid == 0) s_grp = atomicAdd(group_counter, 1ull);
        __syncthreads();
        const int64_t grp = (int64_t)s_grp;
        if (grp >= ngroups) break;
        const int64_t t0g = grp * PA_GROUP;
        const int ng = (int)((ntasks < t0g + PA_GROUP ? ntasks : t0g + PA_GROUP) - t0g);
        if (seg_by_group) { // nseg = ranges x PA_WAVES: the range of the group, this wavefront's own counter inside it
            const int gseg = (int)(grp * (int64_t)(nseg / PA_WAVES) / ngroups) * PA_WAVES + wave;
            if (gseg != seg) {
                flush(); // what is staged belongs to the previous group's segment
                seg = gseg;
                seg_list = cand + (int64_t)seg * seg_cap;
            }
        }
        __syncthreads(); // the previous group is finished with the tables
        if (tid < ng) {
            const Task t = tasks[t0g + tid];
            const int n = nvalid[t.q];
            s_q[tid] = (int32_t)t.q;
            s_rc[tid] = t.rc;
            s_tb[tid] = t.tBegin;
            s_wlen[tid] = t.wlen;
            s_npos[tid] = n > 0 && t.wlen - K + 1 > 0 ? t.wlen - K + 1 : 0;
            s_goff[tid] = t.g >= 0 ? ix.g_off[t.g] : -1;
            s_woff[tid] = t.woff;
            s_bits[tid] = cmp_bits ? bits_off[t.q] : -1;
            s_log[tid] = cmp_bits ? bits_log[t.q] : 0;
        }
        __syncthreads();
        int r0 = 0;
        while (r0 < ng) { // runs of tasks of one query (nearly always the whole group)
            int r1 = r0 + 1;
            while (r1 < ng && s_q[r1] == s_q[r0]) r1++;
            const int log = s_log[r0], blog = lm_pa_bloom_log(log);
            const uint32_t *qbits = s_bits[r0] >= 0 ? cmp_bits + s_bits[r0] : nullptr;
            cur_qbits = qbits;
            cur_log = log;
            cur_blog = blog;
            // this query's LDS maps (K >= 16 and a query with maps: the fast path exists for some window of it)
            if (qbits != nullptr && K >= 16) {
                const uint32_t *gbl = qbits + lm_pa_bloom_word0(log), *g9 = qbits + lm_pa_map9_word0(log);
                const int nb = 1 << (blog - 5);
                for (int j = tid; j < nb; j += PA_THREADS) s_bloom[j] = gbl[j];
                for (int j = tid; j < (1 << (LM_PA_MAP9_LOG - 5)); j += PA_THREADS) s_map9[j] = g9[j];
            }
            if (tid == 0) {
                int acc = 0;
                for (int j = r0; j < r1; j++) {
                    s_first[j] = acc;
                    acc += (s_npos[j] + (ROLL ? PA_SLICE_ROLL : PA_SLICE) - 1) / (ROLL ? PA_SLICE_ROLL : PA_SLICE);
                }
                s_first[r1] = acc;
                s_next = 0;
            }
            __syncthreads();
            const int nslices = s_first[r1];
            while (true) {
                int sl = 0;
                if (lane == 0) sl = atomicAdd(&s_next, 1);
                sl = __builtin_amdgcn_readfirstlane(sl);
                if (sl >= nslices) break;
                int lo = r0, hi = r1; // the task of slice sl: last j with s_first[j] <= sl
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (s_first[mid] <= sl) lo = mid; else hi = mid;
                }
                const int j = lo;
                Task t; // the fields pa_prefixes / pa_kmer read
                t.rc = s_rc[j];
                t.tBegin = s_tb[j];
                t.wlen = s_wlen[j];
                const int npos = s_npos[j];
                // (ROLL: the slices of a window are equal - 64 lanes x the same number of positions, none nearly empty)
                const int slice = ROLL ? (npos + (s_first[j + 1] - s_first[j]) - 1) / (s_first[j + 1] - s_first[j]) : PA_SLICE;
                const int p0 = (sl - s_first[j]) * slice, p1 = p0 + slice < npos ? p0 + slice : npos;
                const int64_t goff = s_goff[j];
                const uint8_t *gb = goff >= 0 ? ix.gbits : nullptr;
                const uint8_t *w = wbuf + s_woff[j];
                const int m = pa_min_prefix(min_prefix, t.wlen);
                const int p = m > K ? K : m;
                const bool use_bits = qbits != nullptr && p >= LM_PFX_BASES && K >= LM_PFX_BASES;
                const bool fast_pfx = use_bits && gb != nullptr && p <= 15 && K >= 16;
                const uint64_t rec_t = (uint64_t)(t0g + j) << 32;
                if (ROLL && fast_pfx && K == 31 && p1 > p0) {
                    const int np = p1 - p0;
                    const int bpl = (np + 63) >> 6;                                          // positions per lane, <= 32
                    const int g0 = t.rc ? t.tBegin + t.wlen - K - (p1 - 1) : t.tBegin + p0;  // first genome position
                    const int64_t abs0 = goff * 4 + g0;                                      // in bases from gbits
                    const int rl0 = bpl * lane;                                              // the lane's first position (genome order)
                    const int64_t ab = abs0 + (rl0 < np ? rl0 : 0);
                    // the lane's string: 64 bases from its first position = dwords S0..S3 (first base in the top bits), cut
                    // out of five consecutive dwords of the 2-bit genome; dwords past the last base the slice needs are not
                    // read (the store is padded by one 8-byte word only): their bits belong to positions >= np
                    const uint32_t *gd = (const uint32_t *)gb;
                    const int64_t d0 = ab >> 4, dlast = (abs0 + np + K - 2) >> 4;
                    uint32_t e[5];
#pragma unroll
                    for (int k = 0; k < 5; k++) e[k] = __builtin_bswap32(gd[d0 + k < dlast ? d0 + k : dlast]);
                    const int ob = (int)(ab & 15) * 2;
                    uint32_t S0, S1, S2, S3;
                    {
                        const uint64_t q0_ = ((uint64_t)e[0] << 32) | e[1], q1_ = ((uint64_t)e[1] << 32) | e[2];
                        const uint64_t q2_ = ((uint64_t)e[2] << 32) | e[3], q3_ = ((uint64_t)e[3] << 32) | e[4];
                        S0 = (uint32_t)(q0_ >> (32 - ob));
                        S1 = (uint32_t)(q1_ >> (32 - ob));
                        S2 = (uint32_t)(q2_ >> (32 - ob));
                        S3 = (uint32_t)(q3_ >> (32 - ob));
                    }
                    // its reverse complement: base i of R = the complement of base 63 - i of S.  The reverse-complement k-mer of
                    // position j (bases [j, j + 31) of S) starts at base 33 - j of R.
                    uint32_t R0 = pa_rc16(S3), R1 = pa_rc16(S2), R2 = pa_rc16(S1), R3 = pa_rc16(S0);
                    const int shp = 32 - 2 * p, sha = (p - LM_PFX_BASES) << 1;
                    const uint32_t m9p = (1u << ((p - 9) << 1)) - 1u;
                    const uint32_t pcode = (uint32_t)((p - LM_PFX_BASES) >> 1) << 30; // p = 11, 13 or 15
#pragma unroll 1
                    for (int half = 0; half < 2; half++) {
                        // positions 16 * half + jj of the lane: windows over (S0, S1) and (R1, R2, R3) with immediate shifts
#pragma unroll
                        for (int jj = 0; jj < 16; jj++) {
                            if (16 * half + jj >= bpl) break; // (wave-uniform)
                            const int sh = 2 * jj;            // first p bases of the k-mer: bit offset sh of (S0, S1)
                            const uint32_t xw = sh ? __builtin_amdgcn_alignbit(S0, S1, 32 - sh) : S0;
                            const int orc = 66 - 2 * jj;      // ... of its reverse complement: bit offset orc of (R0..R3)
                            const uint32_t Ra = (orc >> 5) == 2 ? R2 : R1, Rb = (orc >> 5) == 2 ? R3 : R2;
                            const uint32_t yw = (orc & 31) ? __builtin_amdgcn_alignbit(Ra, Rb, 32 - (orc & 31)) : Ra;
                            const uint32_t first = xw >> shp, rl = yw >> shp;
                            const uint32_t pf0 = t.rc ? rl : first, pf1 = t.rc ? first : rl;
                            const int r = rl0 + 16 * half + jj;           // position of the slice in genome order
                            const int i = t.rc ? p1 - 1 - r : p0 + r;    // window position
                            const bool in = r < np;
                            bool c0, c1;
                            const uint32_t a0 = pf0 >> sha, a1 = pf1 >> sha;
                            const uint32_t s00 = lm_pa_bloom_slot(a0, 0, blog), s01 = lm_pa_bloom_slot(a0, 1, blog);
                            const uint32_t s10 = lm_pa_bloom_slot(a1, 0, blog), s11 = lm_pa_bloom_slot(a1, 1, blog);
                            const uint32_t b00 = s_bloom[s00 >> 5], b01 = s_bloom[s01 >> 5];
                            const uint32_t b10 = s_bloom[s10 >> 5], b11 = s_bloom[s11 >> 5];
                            const bool h0 = in && (((b00 >> (s00 & 31)) & (b01 >> (s01 & 31))) & 1u) != 0;
                            const bool h1 = in && (((b10 >> (s10 & 31)) & (b11 >> (s11 & 31))) & 1u) != 0;
                            const bool q0 = in && !h0 && (pf0 & m9p) == 0, q1 = in && !h1 && (pf1 & m9p) == 0;
                            c0 = h0;
                            c1 = h1;
                            if (log > blog) { // long reads: the global 11-base bitmap, 64 pending positions at a time (see below)
                                c0 = c1 = false;
                                const uint64_t rec = rec_t | ((uint64_t)(uint32_t)i << 1);
                                pend(h0 || q0, pf0 | pcode, rec);
                                pend(h1 || q1, pf1 | pcode, rec | 1ull);
                            } else if (__ballot(q0 || q1) != 0ull) { // the partial-prefix rule, all in LDS here (rare lanes)
                                c0 = pa_partial_rule(q0, c0, pf0, p, blog, s_bloom, s_map9); // (lm_pa_candidate2 for these lanes, in LDS)
                                c1 = pa_partial_rule(q1, c1, pf1, p, blog, s_bloom, s_map9);
                            }
                            const uint64_t m0 = __ballot(c0), m1 = __ballot(c1);
                            if ((m0 | m1) != 0ull) {
                                const int n0 = __popcll(m0);
                                const uint64_t rec = rec_t | ((uint64_t)(uint32_t)i << 1);
                                if (c0) stg[n_stg + __popcll(m0 & lt_mask)] = rec;
                                if (c1) stg[n_stg + n0 + __popcll(m1 & lt_mask)] = rec | 1ull;
                                n_stg += n0 + __popcll(m1);
                                if (n_stg > PA_STAGE - 128) flush(); // LDS accesses of one wavefront complete in program order
                            }
                        }
                        S0 = S1; // the second half: the same windows one dword further (and one dword earlier in R)
                        S1 = S2;
                        R3 = R2;
                        R2 = R1;
                        R1 = R0;
                    }
                } else if (fast_pfx) {
                    // prefixes only; the low-complexity filter (which needs the whole k-mer) is applied by k_pa_search.
                    // The 2-bit genome words of the whole slice in ONE load (lane l holds 32 bases: 64 lanes cover the
                    // slice's 1920 positions + K - 1 + the word misalignment); a position takes its two words from the
                    // lanes that hold them.  Positions are walked in genome order (a reverse-strand window runs backwards).
                    const int np = p1 - p0;
                    const int g0 = t.rc ? t.tBegin + t.wlen - K - (p1 - 1) : t.tBegin + p0; // first genome position
                    const int64_t abs0 = goff * 4 + g0;                                      // in bases from gbits
                    const int64_t w0 = abs0 >> 5;
                    const int nwords = (int)(((abs0 + np + K - 2) >> 5) - w0) + 2; // + the word the funnel shift reads
                    const int lw = lane < nwords ? lane : nwords - 1;
                    const uint64_t Wl = __builtin_bswap64(((const uint64_t *)gb)[w0 + lw]);
                    for (int tile = 0; tile < np; tile += 64) {
                        const int r = tile + lane; // position of the slice in genome order
                        const int64_t ab = abs0 + r;
                        const int wi = (int)((ab >> 5) - w0), o = (int)(ab & 31) * 2;
                        const uint64_t H = __shfl(Wl, wi, 64), L = __shfl(Wl, wi + 1, 64);
                        const uint64_t v = o ? ((H << o) | (L >> (64 - o))) : H; // 32 bases from the position, left aligned
                        const uint32_t first = (uint32_t)(v >> (64 - 2 * p));
                        const uint32_t last = (uint32_t)(v >> (64 - 2 * K)) & ((1u << (2 * p)) - 1u);
                        const uint32_t rl = revcomp_small(last, p);
                        const uint32_t pf0 = t.rc ? rl : first, pf1 = t.rc ? first : rl;
                        const int i = t.rc ? p1 - 1 - r : p0 + r; // window position
                        const bool in = r < np;
                        // lm_pa_candidate2 for both strands with the common case free of branches: two Bloom bits per
                        // strand from LDS; the global 11-base bitmap (long reads) and the partial-prefix rule (bases [9, p)
                        // all A: one k-mer in 16) are left to the few lanes that need them
                        bool c0, c1;
                        {
                            const uint32_t a0 = pf0 >> ((p - LM_PFX_BASES) << 1), a1 = pf1 >> ((p - LM_PFX_BASES) << 1);
                            const uint32_t s00 = lm_pa_bloom_slot(a0, 0, blog), s01 = lm_pa_bloom_slot(a0, 1, blog);
                            const uint32_t s10 = lm_pa_bloom_slot(a1, 0, blog), s11 = lm_pa_bloom_slot(a1, 1, blog);
                            const uint32_t b00 = s_bloom[s00 >> 5], b01 = s_bloom[s01 >> 5];
                            const uint32_t b10 = s_bloom[s10 >> 5], b11 = s_bloom[s11 >> 5];
                            const bool h0 = in && (((b00 >> (s00 & 31)) & (b01 >> (s01 & 31))) & 1u) != 0;
                            const bool h1 = in && (((b10 >> (s10 & 31)) & (b11 >> (s11 & 31))) & 1u) != 0;
                            const uint32_t m9p = (1u << ((p - 9) << 1)) - 1u;
                            const bool q0 = in && !h0 && (pf0 & m9p) == 0, q1 = in && !h1 && (pf1 & m9p) == 0;
                            c0 = h0;
                            c1 = h1;
                            if (log > blog) {
                                // long reads: what the Bloom filter lets through must ask the 11-base bitmap in global
                                // memory.  Asked where it arises, nearly every pass of the loop would stall on that load
                                // for the sake of a handful of lanes; the lanes concerned are set aside in the wavefront's
                                // pending list instead and examined 64 at a time (one load for 64 useful lanes)
                                c0 = c1 = false;
                                const uint64_t rec = rec_t | ((uint64_t)(uint32_t)i << 1);
                                const uint32_t pcode = (uint32_t)((p - LM_PFX_BASES) >> 1) << 30; // p = 11, 13 or 15
                                pend(h0 || q0, pf0 | pcode, rec);
                                pend(h1 || q1, pf1 | pcode, rec | 1ull);
                            } else if (__ballot(q0 || q1) != 0ull) {
                                // the partial-prefix rule, all in LDS here (rare lanes)
                                c0 = pa_partial_rule(q0, c0, pf0, p, blog, s_bloom, s_map9); // (lm_pa_candidate2 for these lanes, in LDS)
                                c1 = pa_partial_rule(q1, c1, pf1, p, blog, s_bloom, s_map9);
                            }
                        }
                        // both strands appended with one reservation in the wavefront's strip
                        const uint64_t m0 = __ballot(c0), m1 = __ballot(c1);
                        if ((m0 | m1) != 0ull) {
                            const int n0 = __popcll(m0);
                            const uint64_t rec = rec_t | ((uint64_t)(uint32_t)i << 1);
                            if (c0) stg[n_stg + __popcll(m0 & lt_mask)] = rec;
                            if (c1) stg[n_stg + n0 + __popcll(m1 & lt_mask)] = rec | 1ull;
                            n_stg += n0 + __popcll(m1);
                            if (n_stg > PA_STAGE - 128) flush(); // LDS accesses of one wavefront complete in program order
                        }
                    }
                } else {
                    for (int tile = p0; tile < p1; tile += 64) {
                        const int i = tile + lane;
                        uint64_t kmer = 0, rc = 0;
                        if (i < p1) pa_kmer(t, w, gb, goff, i, K, &kmer, &rc);
#pragma unroll
                        for (int strand = 0; strand < 2; strand++) {
                            const bool c = i < p1 && (!use_bits || lm_pa_candidate(qbits, log, strand ? rc : kmer, p, K));
                            push(c, rec_t | ((uint64_t)(uint32_t)i << 1) | (uint64_t)strand);
                        }
                    }
                }
            }
            if (n_pend > 0) pend_round(n_pend); // (n_pend < 64 here)
            __syncthreads(); // everybody is done with this query's maps and the slice table
            r0 = r1;
        }
    }
    flush();
}

#define PAS_STAGE 512 /* anchors a wavefront of k_pa_search stages in LDS between two appends to the global list */
__global__ __launch_bounds__(256) void k_pa_search(DevIndexView ix, const Task *__restrict__ tasks,
                                                    const uint8_t *__restrict__ wbuf,
                                                    const uint64_t *__restrict__ keys_cmp,
                                                    const uint32_t *__restrict__ vals_cmp,
                                                    const int64_t *__restrict__ posoff, const int32_t *__restrict__ nvalid,
                                                    const uint32_t *__restrict__ cmp_tab,
                                                    const int64_t *__restrict__ tab_off,
                                                    const int32_t *__restrict__ tab_bits, int K, int min_prefix,
                                                    const unsigned long long *__restrict__ seg_count, int64_t seg_cap,
                                                    int blocks_per_seg, const uint64_t *__restrict__ cand_all,
                                                    unsigned long long *__restrict__ count, int64_t cap,
                                                    uint64_t *__restrict__ outA, uint64_t *__restrict__ outB, int qbits,
                                                    int tbits, int nseg, int xcd_map, unsigned long long *__restrict__ dbg) {
    // qbits > 0: compact single-key anchors, task | QBegin:qbits | (32-Len):6 | TBegin:tbits | 2 flags in one u64 (outA is
    // not written): same order as (task, B) and one keys-only radix sort over the bits in use instead of two pair sorts.
    // qbits == 0 (batches whose fields need more than 64 bits): (task, B) pairs, the task staged beside B.
    __shared__ uint64_t s_stage[4][PAS_STAGE];
    __shared__ uint32_t s_task[4][PAS_STAGE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t *stg = s_stage[wave];
    uint32_t *stt = s_task[wave];
    int n_stg = 0; // wave-uniform
    const uint64_t lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
    auto flush = [&]() { // this wavefront's staged anchors -> the global list, one atomic
        if (n_stg == 0) return;
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(count, (unsigned long long)n_stg);
        base = __shfl(base, 0, 64);
        for (int j = lane; j < n_stg; j += 64) {
            const unsigned long long idx = base + (unsigned long long)j;
            if ((int64_t)idx < cap) {
                if (qbits == 0) outA[idx] = (uint64_t)stt[j];
                outB[idx] = stg[j];
            }
        }
        n_stg = 0;
    };
    // blocks_per_seg blocks share a segment.  xcd_map: hardware workgroup b runs on XCD b % 8 (round-robin dispatch), so the
    // blocks of a segment are given ids of ONE residue class: the per-query tables its candidates probe (bucket table, sorted
    // k-mers, positions: 1-2 MB) then stay in that XCD's 4-MB L2 instead of being fetched by all eight
    int seg, sub;
    if (xcd_map) { // segments come in ranges of LM_PA_RANGE_SEGS (one range = one or two queries): range R on XCD R % 8
        const int r = blockIdx.x >> 3, sr = r / blocks_per_seg;
        seg = ((sr / LM_PA_RANGE_SEGS) * 8 + (blockIdx.x & 7)) * LM_PA_RANGE_SEGS + sr % LM_PA_RANGE_SEGS;
        sub = r % blocks_per_seg;
        if (seg >= nseg) return;
    } else {
        seg = blockIdx.x / blocks_per_seg;
        sub = blockIdx.x % blocks_per_seg;
    }
    const uint64_t *cand = cand_all + (int64_t)seg * seg_cap;
    int64_t nc = (int64_t)seg_count[seg];
    if (nc > seg_cap) nc = seg_cap; // the host re-runs both kernels with a larger list
    const int sh_t = 2, sh_l = 2 + tbits, sh_q = 8 + tbits, sh_a = 8 + tbits + qbits;
    const uint64_t ccc = lm_ns(1, K), ggg = lm_ns(2, K), ttt = lm_kmer_mask(K);
    const int64_t stride = (int64_t)blocks_per_seg * blockDim.x;
    for (int64_t base = (int64_t)sub * blockDim.x; base < nc; base += stride) { // whole wavefronts stay together
        const int64_t ci = base + threadIdx.x;
        unsigned long long d_0 = 0, d_1 = 0, d_it = 0, d_an = 0;
        if (dbg) d_0 = wall_clock64();
        int j = 0, hi = 0, i = 0;
        bool rcs = false;
        uint64_t key = 0, right = 0;
        int64_t ti = 0;
        uint32_t qb = 0, qe = 0;
        const uint64_t *keys = nullptr;
        const uint32_t *vals = nullptr;
        if (ci < nc) {
            const uint64_t rec = cand[ci];
            ti = (int64_t)(rec >> 32);
            i = (int)((uint32_t)rec >> 1);
            rcs = (rec & 1ull) != 0;
            const Task t = tasks[ti];
            const uint8_t *gb = t.g >= 0 ? ix.gbits : nullptr;
            const int64_t goff = t.g >= 0 ? ix.g_off[t.g] : 0;
            uint64_t kmer, rc;
            pa_kmer(t, wbuf + t.woff, gb, goff, i, K, &kmer, &rc);
            key = rcs ? rc : kmer;
            keys = keys_cmp + 2 * posoff[t.q];
            vals = vals_cmp + 2 * posoff[t.q];
            qb = (uint32_t)t.qBegin;
            qe = (uint32_t)t.qEnd;
            const bool lowc = kmer == 0 || kmer == ccc || kmer == ggg || kmer == ttt; // lib-seq_compare.go:374
            if (lowc || !lm_tree_search_first_tab(keys, nvalid[t.q], key, pa_min_prefix(min_prefix, t.wlen), K,
                                                  cmp_tab + tab_off[t.q], tab_bits[t.q], &j, &hi,
                                                  &right))
                j = hi = 0;
        }
        // the matches of all lanes, one per lane and round, appended to the wavefront's LDS strip
        if (dbg) d_1 = wall_clock64();
        while (true) {
            d_it++;
            bool live = j < hi;
            uint64_t kj = 0;
            if (live) {
                kj = keys[j];
                live = kj <= right;
                if (!live) hi = j; // past the keys sharing p bases
            }
            if (__ballot(live) == 0) break;
            bool ok = false;
            uint64_t B = 0;
            if (live) {
                const uint32_t v = vals[j];
                const uint32_t lp = (uint32_t)lm_lcp(kj, key, K);
                if (!rcs) {
                    const uint32_t pp = v >> 1;
                    if (!((v & 1u) == 1u || pp < qb || pp + lp > qe)) {
                        ok = true;
                        B = lm_pack_anchor((int)pp, (int)lp, i, false, false);
                    }
                } else {
                    const uint32_t pp = (v >> 1) + (uint32_t)K - lp;
                    if (!((v & 1u) == 0u || pp + lp < qb || pp > qe)) {
                        ok = true;
                        B = lm_pack_anchor((int)pp, (int)lp, i + K - (int)lp, true, true);
                    }
                }
                j++;
            }
            const uint64_t m = __ballot(ok);
            if (m) {
                if (ok) {
                    if (qbits > 0) { // repack the fields of B under the task number
                        const LmSub u = lm_unpack_anchor(B);
                        B = ((uint64_t)ti << sh_a) | ((uint64_t)(uint32_t)u.qbegin << sh_q) |
                            ((uint64_t)(32 - (int)u.len) << sh_l) | ((uint64_t)(uint32_t)u.tbegin << sh_t) |
                            ((uint64_t)u.qrc << 1) | (uint64_t)u.trc;
                    }
                    const int slot = n_stg + __popcll(m & lt_mask);
                    stg[slot] = B;
                    stt[slot] = (uint32_t)ti;
                }
                n_stg += __popcll(m);
                d_an += (unsigned long long)__popcll(m);
                if (n_stg > PAS_STAGE - 64) flush(); // LDS accesses of one wavefront complete in program order
            }
        }
        if (dbg) { // LM_DEBUG_PA_SEARCH: {candidates, wavefront passes, enumeration passes, anchors, clocks of the search, of the enumeration}
            const unsigned long long nc_w = (unsigned long long)__popcll(__ballot(ci < nc)), d_2 = wall_clock64();
            if (lane == 0) {
                atomicAdd(dbg + 0, nc_w);
                atomicAdd(dbg + 1, 1ull);
                atomicAdd(dbg + 2, d_it);
                atomicAdd(dbg + 3, d_an);
                atomicAdd(dbg + 4, d_1 - d_0);
                atomicAdd(dbg + 5, d_2 - d_1);
            }
        }
    }
    flush();
}
// first anchor of every task in the (task, B)-sorted list
__global__ void k_pa_task_off_sorted(const uint64_t *__restrict__ sortedA, int shift, int64_t total, int64_t ntasks,
                                     int64_t *__restrict__ pa_off) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= ntasks; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t lo = 0, hi = total;
        while (lo < hi) {
            int64_t mid = (lo + hi) >> 1;
            if ((sortedA[mid] >> shift) < (uint64_t)i)
                lo = mid + 1;
            else
                hi = mid;
        }
        pa_off[i] = lo;
    }
}


// Clear + Trim + Chainer2 per chain (lib-seq_compare.go:447-508)


// ------------------------------------------------------------------------------------------------------------
// Wave-cooperative Clear + Trim + Chainer2 (+ chainARegion) for one chain: lanes cover anchors for ClearSubstrPairs and
// candidate predecessors j for the banded DP; emission order and every tie rule are those of lm_clear_sorted / lm_trim
// / lm_run_chain2 (lm_algos.h), which stay the CPU-checked statement of the logic.
#include "lm_pa_chain_dp.h"
#include "lm_pa_chain_bt.h"

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long x = __shfl_xor(v, o, 64);
        v = x > v ? x : v;
    }
    return v;
}

// (Forms measured against this one on one resident index and removed in round 5: the DP through global memory, 152 -> 130 ms per
// C4 launch; the backtrack by lane 0 and ClearSubstrPairs by binary search in global memory, C3 9.85 -> 10.0 s, C4 shard 1.47 ->
// 1.56 s per step; the DP of long windows pipelined over the eight wavefronts of a workgroup, C4 shard 1.47 vs 1.44 s without.)
// (held to 64 VGPRs = 8 wavefronts per SIMD - the kernel is bound by the latency of a wavefront's dependent chain x windows in
// flight; 67 VGPRs were 7: C4 shard 1.40 -> 1.33 s per step)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8))) void k_pa_chain_wave(const uint64_t *__restrict__ B, const int64_t *__restrict__ pa_off,
                                                       int64_t ntasks, int K, LmChain2Opt opt, LmSub *__restrict__ subs_pool,
                                                       uint8_t *__restrict__ marks_pool, uint64_t *__restrict__ msi_pool,
                                                       int32_t *__restrict__ stack_pool, LmChain2 *__restrict__ out_pool,
                                                       int32_t *__restrict__ out_n, int32_t *__restrict__ clr_n, int qbits,
                                                       int tbits,
                                                       unsigned long long *__restrict__ dbg) {
    const int lane = threadIdx.x;
    __shared__ PcdLds pcd_lds;
    __shared__ PcbLds pcb_lds;
    for (int64_t ti = blockIdx.x; ti < ntasks; ti += gridDim.x) {
        const int64_t o = pa_off[ti];
        int n = (int)(pa_off[ti + 1] - o);
        __syncthreads();
        if (n <= 0) {
            if (lane == 0) {
                out_n[ti] = 0;
                clr_n[ti] = 0;
            }
            continue;
        }
        LmSub *sb = subs_pool + o;
        uint8_t *marks = marks_pool + o;
        uint64_t *msi = msi_pool + o;
        LmChain2 *res = out_pool + o;
        unsigned long long d_0 = 0, d_1 = 0, d_2 = 0;
        if (dbg) d_0 = wall_clock64();
        for (int i = lane; i < n; i += 64) {
            const uint64_t v = B[o + i];
            if (qbits > 0) { // compact single-key form (see k_pa_search)
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
        __syncthreads();
        // ---- ClearSubstrPairs (lib-index-search.go:927-972): anchor i+1 is dropped when nested in an earlier one ----
        if (n > 1) {
            pa_clear_marks_wave(sb, n, K, marks, (PccLds *)&pcd_lds); // (lm_pa_clear_tile.h)
            __syncthreads();
            int w = 0; // ordered in-place compaction, chunk by chunk
            for (int c = 0; c < n; c += 64) {
                int i = c + lane;
                bool keep = i < n && !marks[i];
                LmSub v;
                if (keep) v = sb[i];
                unsigned long long bal = __ballot(keep);
                int before = __popcll(bal & ((1ull << lane) - 1ull));
                __syncthreads();
                if (keep) sb[w + before] = v;
                w += __popcll(bal);
            }
            n = w;
            __syncthreads();
        }
        // ---- TrimSubStrPairs (lane 0; it stops after a few anchors) ----
        int start = 0;
        if (lane == 0) n = lm_trim(sb, n, 100.0f, &start);
        n = __shfl(n, 0, 64);
        start = __shfl(start, 0, 64);
        if (lane == 0) clr_n[ti] = n;
        if (n <= 0) {
            if (lane == 0) out_n[ti] = 0;
            continue;
        }
        const LmSub *a_ = sb + start;
        if (n == 1) {
            if (lane == 0) out_n[ti] = lm_run_chain2(a_, 1, opt, msi, stack_pool + 2 * o + 4 * ti, res);
            continue;
        }
        // ---- banded DP (lib-chaining2.go:222-307), candidates j scanned 64 at a time from i-1 downwards ----
        if (dbg) d_1 = wall_clock64();
        long long M = 0;
        int Mi = 0;
        pa_chain_dp_reg(a_, n, opt, msi, &pcd_lds, &M, &Mi); // (lm_pa_chain_dp_core.h: the last 64 anchors in registers)
        __threadfence_block();
        __syncthreads();
        // ---- backtrack by the wavefront: region scans by 64 lanes, the walk out of LDS tiles (lm_pa_chain_bt.h) ----
        if (dbg) d_2 = wall_clock64();
        {
            const int no = pa_chain_backtrack_wave(a_, n, opt, msi, M, Mi, stack_pool + 2 * o + 4 * ti, res, &pcb_lds);
            if (lane == 0) out_n[ti] = no;
        }
        if (dbg && lane == 0) {
            const unsigned long long d_3 = wall_clock64();
            atomicAdd(dbg + 0, d_1 - d_0);
            atomicAdd(dbg + 1, d_2 - d_1);
            atomicAdd(dbg + 2, d_3 - d_2);
            atomicAdd(dbg + 3, 1ull);
            const int cls = n <= 8 ? 0 : (n <= 64 ? 1 : (n <= 256 ? 2 : 3)); // by anchors left after clear + trim: windows, anchors, clocks
            atomicAdd(dbg + 4 + 3 * cls, 1ull);
            atomicAdd(dbg + 5 + 3 * cls, (unsigned long long)n);
            atomicAdd(dbg + 6 + 3 * cls, d_3 - d_0);
        }
    }
}


__global__ void k_gather_chain2(const LmChain2 *__restrict__ in, const int64_t *__restrict__ pa_off,
                                const int32_t *__restrict__ out_n, const int64_t *__restrict__ res_off, int64_t ntasks,
                                LmChain2 *__restrict__ out) {
    for (int64_t ti = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ti < ntasks; ti += (int64_t)gridDim.x * blockDim.x) {
        const LmChain2 *src = in + pa_off[ti];
        LmChain2 *dst = out + res_off[ti];
        for (int i = 0; i < out_n[ti]; i++) dst[i] = src[i];
    }
}

// ------------------------------------------------------------------------------------------------------------
// extendMatch (lib-index-search-util.go:34-96) — scratch sizing pass + run pass
__device__ __forceinline__ void ext_flanks(const HspIn &h, int len1, int len2, int *r_n1, int *r_n2, int *l_n1, int *l_n2) {
    *r_n1 = *r_n2 = *l_n1 = *l_n2 = 0;
    const int m = 2;
    if (h.end1 + m < len1 && h.end2 + m < len2) {
        int e = h.rc ? (h.ext_len < h.tbegin ? h.ext_len : h.tbegin) : (h.ext_len < h.max_ext_len ? h.ext_len : h.max_ext_len);
        if (e > 2) {
            *r_n1 = (h.end1 + e < len1 ? h.end1 + e : len1) - h.end1;
            *r_n2 = (h.end2 + e < len2 ? h.end2 + e : len2) - h.end2;
        }
    }
    if (h.start1 > m && h.start2 > m) {
        int e = h.rc ? (h.ext_len < h.max_ext_len ? h.ext_len : h.max_ext_len) : (h.ext_len < h.tbegin ? h.ext_len : h.tbegin);
        if (e > 2) {
            *l_n1 = h.start1 - (h.start1 - e > 0 ? h.start1 - e : 0);
            *l_n2 = h.start2 - (h.start2 - e > 0 ? h.start2 - e : 0);
        }
    }
}

__device__ __forceinline__ int count_2mer_pairs(const uint8_t *a, int n1, const uint8_t *b, int n2) {
    if (n1 < 2 || n2 < 2) return 0;
    int c1[16], c2[16];
    for (int i = 0; i < 16; i++) c1[i] = c2[i] = 0;
    for (int i = 0; i + 1 < n1; i++) c1[(lm_base2bit(a[i]) << 2) | lm_base2bit(a[i + 1])]++;
    for (int i = 0; i + 1 < n2; i++) c2[(lm_base2bit(b[i]) << 2) | lm_base2bit(b[i + 1])]++;
    int s = 0;
    for (int i = 0; i < 16; i++) s += c1[i] * c2[i];
    return s;
}

__global__ void k_extend_count(const HspIn *__restrict__ hsps, int64_t n, const uint8_t *__restrict__ qseq,
                               const int64_t *__restrict__ qoff, const uint8_t *__restrict__ wbuf,
                               int32_t *__restrict__ cap) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const HspIn h = hsps[i];
        const uint8_t *s1 = qseq + qoff[h.q];
        const uint8_t *s2 = wbuf + h.woff;
        int rn1, rn2, ln1, ln2;
        ext_flanks(h, h.len1, h.len2, &rn1, &rn2, &ln1, &ln2);
        int a = count_2mer_pairs(s1 + h.end1, rn1, s2 + h.end2, rn2);
        int b = count_2mer_pairs(s1 + h.start1 - ln1, ln1, s2 + h.start2 - ln2, ln2); // reversal keeps 2-mer pair counts
        cap[i] = (a > b ? a : b) + 1;
    }
}

// One flank of extendMatch: lm_extend_flank_grid (lm_algos.h) - bit-parallel 2-mer pairs + Chainer3 on the (q, t) grid,
// with the per-item scratch TRANSPOSED across the wavefront (element j of lane L at [j * 64 + L]) so the 64 concurrent
// chainers of a wave share cache lines.
// scratch rows needed by each wavefront of k_extend: the largest pair count of its 32 HSPs
__global__ void k_extend_wave_cap(const int32_t *__restrict__ cap, int64_t n, int32_t *__restrict__ wcap, int64_t nw) {
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nw; w += (int64_t)gridDim.x * blockDim.x) {
        int m = 0;
        for (int64_t i = 32 * w; i < 32 * w + 32 && i < n; i++) m = cap[i] > m ? cap[i] : m;
        wcap[w] = m;
    }
}

// work item = (HSP, side): the two flanks of an HSP are independent, which doubles the parallelism of this
// latency-bound stage. side 0 = right flank -> (e1, e2), side 1 = left flank -> (s1, s2). One wavefront per workgroup;
// woff[w] = first scratch row of wavefront w.
__global__ __launch_bounds__(64) void k_extend(const HspIn *__restrict__ hsps, int64_t n, const uint8_t *__restrict__ qseq,
                                               const int64_t *__restrict__ qoff, const uint8_t *__restrict__ wbuf,
                                               const int32_t *__restrict__ cap, const int64_t *__restrict__ woff,
                                               uint16_t *__restrict__ subs, int32_t *__restrict__ msi,
                                               LmM128 *__restrict__ rows_pool, uint32_t *__restrict__ rstart_pool,
                                               HspExt *__restrict__ out) {
    // grid-chainer rows: one private [LM_EXT_ROWS][64] slab per workgroup (the grid is sized to the resident wavefronts)
    LmM128 *rows = rows_pool + (int64_t)blockIdx.x * LM_EXT_ROWS * 64 + threadIdx.x;
    uint32_t *rstart = rstart_pool + (int64_t)blockIdx.x * LM_EXT_ROWS * 64 + threadIdx.x;
    for (int64_t wv = blockIdx.x; wv * 64 < 2 * n; wv += gridDim.x) {
        const int64_t w = wv * 64 + threadIdx.x;
        if (w >= 2 * n) continue;
        const int64_t i = w >> 1;
        const int side = (int)(w & 1);
        const HspIn h = hsps[i];
        const uint8_t *seq1 = qseq + qoff[h.q];
        const uint8_t *seq2 = wbuf + h.woff;
        const bool rc = h.rc != 0;
        const int m = 2;
        uint16_t *sb = subs + woff[wv] * 64 + threadIdx.x;
        int32_t *ms = msi + woff[wv] * 64 + threadIdx.x;
        int d1 = 0, d2 = 0;
        if (side == 0) {
            if (h.end1 + m < h.len1 && h.end2 + m < h.len2) {
                int ext = rc ? (h.ext_len < h.tbegin ? h.ext_len : h.tbegin)
                             : (h.ext_len < h.max_ext_len ? h.ext_len : h.max_ext_len);
                if (ext > 2) {
                    int e1 = h.end1 + ext < h.len1 ? h.end1 + ext : h.len1;
                    int e2 = h.end2 + ext < h.len2 ? h.end2 + ext : h.len2;
                    lm_extend_flank_grid(seq1 + h.end1, e1 - h.end1, seq2 + h.end2, e2 - h.end2, false, sb, ms, cap[i], rows, rstart,
                                         64, &d1, &d2);
                }
            }
            out[i].e1 = d1;
            out[i].e2 = d2;
        } else {
            if (h.start1 > m && h.start2 > m) {
                int ext = rc ? (h.ext_len < h.max_ext_len ? h.ext_len : h.max_ext_len)
                             : (h.ext_len < h.tbegin ? h.ext_len : h.tbegin);
                if (ext > 2) {
                    int s1 = h.start1 - ext > 0 ? h.start1 - ext : 0;
                    int s2 = h.start2 - ext > 0 ? h.start2 - ext : 0;
                    lm_extend_flank_grid(seq1 + s1, h.start1 - s1, seq2 + s2, h.start2 - s2, true, sb, ms, cap[i], rows, rstart, 64,
                                         &d1, &d2);
                }
            }
            out[i].s1 = d1;
            out[i].s2 = d2;
        }
    }
}
// the tail of extendMatch: apply the two flank results and the bounds checks (:84-95)
__global__ void k_extend_fin(const HspIn *__restrict__ hsps, int64_t n, HspExt *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const HspIn h = hsps[i];
        HspExt e = out[i];
        int start1 = h.start1, end1 = h.end1, start2 = h.start2, end2 = h.end2;
        if (e.e1 > 0 || e.e2 > 0) {
            end1 += e.e1;
            end2 += e.e2;
        }
        if (e.s1 > 0 || e.s2 > 0) {
            start1 -= e.s1;
            start2 -= e.s2;
        }
        if (start1 < 0 || start2 < 0) {
            start1 = h.start1;
            start2 = h.start2;
        }
        if (end1 > h.len1 || end2 > h.len2) {
            end1 = h.end1;
            end2 = h.end2;
        }
        e.qs = start1;
        e.qe = end1;
        e.ts = start2;
        e.te = end2;
        out[i] = e;
    }
}

// ------------------------------------------------------------------------------------------------------------
// WFA + BLAST-style score (lib-index-search-util.go:260-304: 2/-3/5/2 over the M-trimmed ops)


// ------------------------------------------------------------------------------------------------------------
// Wave-cooperative WFA: one wavefront (64 lanes) per alignment; lanes own diagonals k = lo + lane (+64 j).
// Same recurrence, trimming, wf-adaptive cut-off and storage layout as lm_wfa_align (lm_algos.h), which is the
// CPU-checked statement of the device logic; the backtrace is the shared lm_wfa_backtrace run by lane 0.
__device__ __forceinline__ int wave_min_i32(int v) {
    // DPP reduction (no LDS round trips: this sits in the score loop of the WFA kernels): pairs, quads, 8 and 16 lanes by
    // rotation inside the rows of 16, then the row totals by the two row broadcasts; the total is in lane 63.
    // All 64 lanes must be active.
    int x;
    x = __builtin_amdgcn_mov_dpp(v, 0xb1, 0xf, 0xf, false); // quad_perm:[1,0,3,2]
    v = x < v ? x : v;
    x = __builtin_amdgcn_mov_dpp(v, 0x4e, 0xf, 0xf, false); // quad_perm:[2,3,0,1]
    v = x < v ? x : v;
    x = __builtin_amdgcn_mov_dpp(v, 0x124, 0xf, 0xf, false); // row_ror:4
    v = x < v ? x : v;
    x = __builtin_amdgcn_mov_dpp(v, 0x128, 0xf, 0xf, false); // row_ror:8
    v = x < v ? x : v;
    x = __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1 and 3
    v = x < v ? x : v;
    x = __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false); // row_bcast:31 into rows 2 and 3
    v = x < v ? x : v;
    return __builtin_amdgcn_readlane(v, 63);
}
// two unsigned 16-bit minima at once (v_pk_min_u16) and their reduction over the wavefront, same DPP steps: the WFA kernels
// keep "first / last diagonal with a property" as (j, W-1-j) pairs, j = diagonal relative to the row's first one
typedef unsigned short lm_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
    lm_u16x2 x = __builtin_bit_cast(lm_u16x2, a), y = __builtin_bit_cast(lm_u16x2, b);
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(x, y));
}
__device__ __forceinline__ uint32_t wave_pkmin_u16(uint32_t v) { // all 64 lanes active; result uniform
    uint32_t x;
    x = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xb1, 0xf, 0xf, false);
    v = pk_min_u16(x, v);
    x = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4e, 0xf, 0xf, false);
    v = pk_min_u16(x, v);
    x = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x124, 0xf, 0xf, false);
    v = pk_min_u16(x, v);
    x = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xf, 0xf, false);
    v = pk_min_u16(x, v);
    x = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xa, 0xf, false);
    v = pk_min_u16(x, v);
    x = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xc, 0xf, false);
    v = pk_min_u16(x, v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ int32_t wf_val(const int32_t *arena, int lo, int hi, int base, int k) {
    return (k < lo || k > hi) ? LM_NULL_OFF : arena[base + (k - lo)];
}

__device__ __forceinline__ int wf_dist(int32_t off, int k, int plen, int tlen) {
    if (off < 0) return 1073741824;
    int lv = plen - (off - k), lh = tlen - off;
    return lv > lh ? lv : lh;
}

// first k in [a, b) (ascending) with pred, else b.  All lanes must call; result uniform.
template <typename F> __device__ __forceinline__ int wave_find_first(int a, int b, int lane, F pred) {
    for (int c = a; c < b; c += 64) {
        int k = c + lane;
        bool p = k < b && pred(k);
        unsigned long long m = __ballot(p);
        if (m) return c + (int)__ffsll((long long)m) - 1;
    }
    return b;
}
// last k in (b, a] (descending from a) with pred, else b.
template <typename F> __device__ __forceinline__ int wave_find_last(int a, int b, int lane, F pred) {
    for (int c = a; c > b; c -= 64) {
        int k = c - lane;
        bool p = k > b && pred(k);
        unsigned long long m = __ballot(p);
        if (m) return c - ((int)__ffsll((long long)m) - 1);
    }
    return b;
}

__device__ __forceinline__ void wave_trim(int32_t *h, const int32_t *arena, int alo, int plen, int tlen, int lane) {
    // h = {lo, hi, base} freshly computed over [lo,hi] with base addressing alo==lo
    int lo = h[0], hi = h[1], base = h[2];
    auto valid = [&](int k) {
        int32_t off = arena[base + (k - alo)];
        return (uint32_t)off <= (uint32_t)tlen && (uint32_t)(off - k) <= (uint32_t)plen;
    };
    int nhi = wave_find_last(hi, lo - 1, lane, valid);     // lo-1 when nothing is valid
    int nlo = wave_find_first(lo, nhi + 1, lane, valid);   // == nhi+1 when empty
    if (nhi < lo) nlo = lo;                                  // sequential code leaves lo untouched in that case
    __syncthreads();
    if (lane == 0) {
        h[2] = base + (nlo - alo);
        h[0] = nlo;
        h[1] = nhi;
    }
}

__global__ __launch_bounds__(64) void k_wfa_wave(const WfaIn *__restrict__ in, int64_t n, const int32_t *__restrict__ todo,
                                                  int64_t ntodo, int32_t *__restrict__ hdr_pool,
                                                  int32_t *__restrict__ arena_pool, uint64_t *__restrict__ ops_pool,
                                                  WfaOut *__restrict__ out) {
    const int lane = threadIdx.x;
    const int X = 4, OE = 8, E = 2;
    for (int64_t x = blockIdx.x; x < ntodo; x += gridDim.x) {
        int64_t i = todo ? todo[x] : x;
        if (i >= n) continue;
        const WfaIn w = in[i];
        const uint8_t *__restrict__ q = w.q;
        const uint8_t *__restrict__ t = w.t;
        const int plen = w.qlen, tlen = w.tlen;
        int32_t *hdr = hdr_pool + w.hdr_off;
        int32_t *arena = arena_pool + w.arena_off;
        const int64_t arena_cap = w.arena_cap;
        const int max_score = w.max_score;
        const int ak = tlen - plen;
        int status = 0;
        int64_t used = 1;
        __syncthreads();
        if (max_score < 1 || arena_cap < 1) {
            status = 1;
        } else if (lane == 0) {
            hdr[0] = 0; hdr[1] = 0; hdr[2] = 0;
            hdr[3] = 1; hdr[4] = -1; hdr[5] = 0;
            hdr[6] = 1; hdr[7] = -1; hdr[8] = 0;
            arena[0] = 0;
        }
        int s = 0;
        while (status == 0) {
            __syncthreads();
            int32_t *hm = hdr + s * 9;
            int mlo = hm[0], mhi = hm[1], mbase = hm[2];
            if (mlo <= mhi) {
                // ---- extend along each diagonal (8 bases per compare) ----
                for (int k = mlo + lane; k <= mhi; k += 64) {
                    int32_t off = arena[mbase + (k - mlo)];
                    if (off < 0) continue;
                    int v = off - k, h = off;
                    while (v + 8 <= plen && h + 8 <= tlen) {
                        uint64_t a, b;
                        __builtin_memcpy(&a, q + v, 8);
                        __builtin_memcpy(&b, t + h, 8);
                        uint64_t d = a ^ b;
                        if (d) {
                            int nb = __builtin_ctzll(d) >> 3;
                            v += nb;
                            h += nb;
                            goto extended;
                        }
                        v += 8;
                        h += 8;
                    }
                    while (v < plen && h < tlen && q[v] == t[h]) {
                        v++;
                        h++;
                    }
                extended:
                    arena[mbase + (k - mlo)] = h;
                }
                __syncthreads();
                if (mlo <= ak && ak <= mhi && arena[mbase + (ak - mlo)] >= tlen) break;
                // ---- wf-adaptive cut-off (min wavefront length 10, max distance diff 50) ----
                int nlo = mlo, nhi = mhi;
                if (mhi - mlo + 1 >= 10) {
                    int dmin = 2147483647;
                    for (int k = mlo + lane; k <= mhi; k += 64) {
                        int d = wf_dist(arena[mbase + (k - mlo)], k, plen, tlen);
                        dmin = d < dmin ? d : dmin;
                    }
                    dmin = wave_min_i32(dmin);
                    auto keep = [&](int k) { return wf_dist(arena[mbase + (k - mlo)], k, plen, tlen) - dmin <= 50; };
                    int top = ak < mhi ? ak : mhi;
                    if (mlo < top) nlo = wave_find_first(mlo, top, lane, keep);
                    int bottom = ak > nlo ? ak : nlo;
                    if (mhi > bottom) nhi = wave_find_last(mhi, bottom, lane, keep);
                }
                __syncthreads();
                if (lane == 0) {
                    hm[2] = mbase + (nlo - mlo);
                    hm[0] = nlo;
                    hm[1] = nhi;
                    for (int c = 1; c <= 2; c++) {
                        int32_t *hc = hm + c * 3;
                        if (hc[0] > hc[1]) continue;
                        if (nlo > hc[0]) {
                            hc[2] += nlo - hc[0];
                            hc[0] = nlo;
                        }
                        if (nhi < hc[1]) hc[1] = nhi;
                    }
                }
            }
            s++;
            if (s >= max_score) {
                status = 1;
                break;
            }
            __syncthreads();
            // ---- compute wavefronts of score s ----
            int32_t *ho = hdr + s * 9;
            int mm_lo = 1, mm_hi = -1, mm_b = 0, mo_lo = 1, mo_hi = -1, mo_b = 0, ie_lo = 1, ie_hi = -1, ie_b = 0, de_lo = 1,
                de_hi = -1, de_b = 0;
            if (s - X >= 0) {
                const int32_t *h = hdr + (s - X) * 9;
                mm_lo = h[0]; mm_hi = h[1]; mm_b = h[2];
            }
            if (s - OE >= 0) {
                const int32_t *h = hdr + (s - OE) * 9;
                mo_lo = h[0]; mo_hi = h[1]; mo_b = h[2];
            }
            if (s - E >= 0) {
                const int32_t *h = hdr + (s - E) * 9;
                ie_lo = h[3]; ie_hi = h[4]; ie_b = h[5];
                de_lo = h[6]; de_hi = h[7]; de_b = h[8];
            }
            int lo = 2147483647, hi = -2147483647;
            bool any = false;
            if (mm_lo <= mm_hi) { any = true; lo = mm_lo < lo ? mm_lo : lo; hi = mm_hi > hi ? mm_hi : hi; }
            if (mo_lo <= mo_hi) { any = true; lo = mo_lo - 1 < lo ? mo_lo - 1 : lo; hi = mo_hi + 1 > hi ? mo_hi + 1 : hi; }
            if (ie_lo <= ie_hi) { any = true; lo = ie_lo + 1 < lo ? ie_lo + 1 : lo; hi = ie_hi + 1 > hi ? ie_hi + 1 : hi; }
            if (de_lo <= de_hi) { any = true; lo = de_lo - 1 < lo ? de_lo - 1 : lo; hi = de_hi - 1 > hi ? de_hi - 1 : hi; }
            if (!any || lo > hi) {
                if (lane == 0) {
                    ho[0] = 1; ho[1] = -1; ho[2] = 0;
                    ho[3] = 1; ho[4] = -1; ho[5] = 0;
                    ho[6] = 1; ho[7] = -1; ho[8] = 0;
                }
                continue;
            }
            int wd = hi - lo + 1;
            if (used + 3ll * wd > arena_cap) {
                status = 1;
                break;
            }
            int bm = (int)used, bi = (int)(used + wd), bd = (int)(used + 2ll * wd);
            used += 3ll * wd;
            for (int k = lo + lane; k <= hi; k += 64) {
                int32_t a = wf_val(arena, mo_lo, mo_hi, mo_b, k - 1), b = wf_val(arena, ie_lo, ie_hi, ie_b, k - 1);
                int32_t ins = (a > b ? a : b) + 1;
                a = wf_val(arena, mo_lo, mo_hi, mo_b, k + 1);
                b = wf_val(arena, de_lo, de_hi, de_b, k + 1);
                int32_t del = a > b ? a : b;
                int32_t mis = wf_val(arena, mm_lo, mm_hi, mm_b, k) + 1;
                int32_t mx = mis > ins ? mis : ins;
                if (del > mx) mx = del;
                uint32_t hh = (uint32_t)mx, vv = (uint32_t)(mx - k);
                if (hh > (uint32_t)tlen) mx = LM_NULL_OFF;
                if (vv > (uint32_t)plen) mx = LM_NULL_OFF;
                arena[bi + (k - lo)] = ins;
                arena[bd + (k - lo)] = del;
                arena[bm + (k - lo)] = mx;
            }
            if (lane == 0) {
                ho[0] = lo; ho[1] = hi; ho[2] = bm;
                ho[3] = lo; ho[4] = hi; ho[5] = bi;
                ho[6] = lo; ho[7] = hi; ho[8] = bd;
            }
            __syncthreads();
            wave_trim(ho + 0, arena, lo, plen, tlen, lane);
            wave_trim(ho + 3, arena, lo, plen, tlen, lane);
            wave_trim(ho + 6, arena, lo, plen, tlen, lane);
        }
        __syncthreads();
        if (lane == 0) {
            WfaOut o;
            o.blast_score = 0;
            if (status != 0) {
                o.r.status = 1;
                o.r.score = 0;
                o.r.nops = 0;
                o.r.qbegin = o.r.qend = o.r.tbegin = o.r.tend = 0;
                o.r.align_len = o.r.matches = o.r.gaps = o.r.gap_regions = 0;
            } else {
                uint64_t *ops = ops_pool + w.ops_off;
                lm_wfa_backtrace(hdr, arena, s, plen, tlen, ops, w.ops_cap, &o.r);
                if (o.r.status == 0) {
                    int first = -1, last = -1;
                    for (int j = 0; j < o.r.nops; j++)
                        if ((ops[j] >> 32) == 'M') {
                            if (first < 0) first = j;
                            last = j;
                        }
                    int score = 0;
                    for (int j = first; j >= 0 && j <= last; j++) {
                        int nn = (int)(ops[j] & 0xffffffffu);
                        char op = (char)(ops[j] >> 32);
                        if (op == 'M')
                            score += nn * 2;
                        else if (op == 'X')
                            score += nn * -3;
                        else
                            score -= 5 + nn * 2;
                    }
                    o.blast_score = score;
                }
            }
            out[i] = o;
        }
    }
}


// ------------------------------------------------------------------------------------------------------------


// 16 bases -> one 32-bit word, first base in the top bits. Any injective 2-bit code works for equality tests:
// (c >> 1) & 3 maps A,C,T,G to 0,1,2,3. *bad is raised for any other byte (the caller falls back to byte compares).
__device__ __forceinline__ uint32_t pack_base(uint32_t c, bool *bad) {
    uint32_t code = (c >> 1) & 3u;
    *bad |= c != ((0x47544341u >> (code << 3)) & 0xffu); // 'A','C','T','G' by code
    return code;
}
__device__ __forceinline__ uint32_t pack16(const uint8_t *__restrict__ s, int nb, bool *bad) {
    uint32_t w = 0;
    if (nb >= 16) {
        uint32_t b[4];
        __builtin_memcpy(b, s, 16);
#pragma unroll
        for (int j = 0; j < 16; j++) w = (w << 2) | pack_base((b[j >> 2] >> ((j & 3) << 3)) & 0xffu, bad);
    } else {
        for (int j = 0; j < nb; j++) w = (w << 2) | pack_base(s[j], bad);
        w <<= 2 * (16 - nb);
    }
    return w;
}
// ---- sliding 2-bit windows of the two sequences of an alignment (k_wfa_lean) --------------------------------------------
// The wavefront kernel used to keep BOTH WHOLE sequences 2-bit packed in LDS: 25 KB for a 50-kb read, which left three to
// six wavefronts per CU (one per SIMD) and nothing to hide the LDS / DPP latencies of the score loop behind.  A wavefront
// only ever reads near its front: under wf-adaptive(10, 50) every kept cell is within 50 + W bases of the leading one.  So
// each sequence is held as a circular window of WFA_WINW words of 16 bases (4096 bases, 1 KB); the LDS of a resident
// wavefront no longer depends on the length of its alignment (and sequences beyond 65 kb need no global-memory fallback).
// Word w of the sequence lives at slot w & (WINW-1); slots 0 and 1 are mirrored behind the last slot so that three
// consecutive words never wrap.  When an extending cell is outside a window, both windows are moved to the SMALLEST
// positions among the extending cells: the cell with the smallest query position is then inside both (two cells of one
// wavefront are less than W < 4000 diagonals apart), so every pass of the extension loop serves somebody; cells further
// ahead wait for the next move.  In the steady state that is one move per ~2900 bases of progress; a cell left far behind
// (possible while a wavefront is narrower than 10 diagonals, before the cut-off applies) costs two moves per score step
// until the cut-off drops it.
#define WFA_WINW 256
struct WfaWin {
    uint32_t *buf;      // WFA_WINW + 2 words of LDS
    const uint8_t *src; // ASCII sequence
    int len;
    int w0;             // words [w0, w0 + WFA_WINW) are resident (wave-uniform)
};
// Makes words [qw0, qw0 + WINW) of Q and [tw0, tw0 + WINW) of T resident (wave-uniform, >= 0); words that stay are not
// reloaded.  ONE loop serves both windows (the byte packing is 60 instructions: the kernel has three copies of this - start
// of an alignment, the extension loop, the replay - instead of two per ring chunk).  *bad: a non-ACGT byte was packed.
__device__ __forceinline__ void wfa_win_move2(WfaWin &Q, int qw0, WfaWin &T, int tw0, int lane, bool *bad, bool fresh) {
    LDS_WAVE_SYNC(); // every lane is done reading the slots that are about to change
    const bool qkeep = !fresh && qw0 >= Q.w0 && qw0 < Q.w0 + WFA_WINW, tkeep = !fresh && tw0 >= T.w0 && tw0 < T.w0 + WFA_WINW;
    const int qfrom = qkeep ? Q.w0 + WFA_WINW : qw0, tfrom = tkeep ? T.w0 + WFA_WINW : tw0;
    const int nq = qw0 + WFA_WINW - qfrom, nt = tw0 + WFA_WINW - tfrom; // words to load (0 when a window does not move)
    for (int i = lane; i < nq + nt; i += 64) {
        const bool isq = i < nq;
        const int w = isq ? qfrom + i : tfrom + (i - nq);
        const uint8_t *src = isq ? Q.src : T.src;
        uint32_t *buf = isq ? Q.buf : T.buf;
        const int nb = (isq ? Q.len : T.len) - 16 * w;
        const uint32_t word = nb > 0 ? pack16(src + 16 * (int64_t)w, nb, bad) : 0u;
        const int slot = w & (WFA_WINW - 1);
        buf[slot] = word;
        if (slot < 2) buf[WFA_WINW + slot] = word;
    }
    Q.w0 = __builtin_amdgcn_readfirstlane(qw0); // provably wave-uniform: the window tests stay cheap
    T.w0 = __builtin_amdgcn_readfirstlane(tw0);
    LDS_WAVE_SYNC();
}
// can 32 bases from `pos` be read from the window ? (three words: pos >> 4 .. +2)
__device__ __forceinline__ bool wfa_win_has(const WfaWin &W, int pos) {
    return (uint32_t)((pos >> 4) - W.w0) < (uint32_t)(WFA_WINW - 2);
}
// 32 packed bases starting at base `pos`, first base in the top bits
__device__ __forceinline__ uint64_t wfa_win_get32(const WfaWin &W, int pos) {
    const uint32_t *p = W.buf + ((pos >> 4) & (WFA_WINW - 1));
    const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
    const int rs = 32 - ((pos & 15) << 1); // 2..32: 64-bit shifts, so that no case needs a branch or a select
    const uint32_t hi = (uint32_t)((((uint64_t)d0 << 32) | d1) >> rs);
    const uint32_t lo = (uint32_t)((((uint64_t)d1 << 32) | d2) >> rs);
    return ((uint64_t)hi << 32) | lo;
}
// 16 packed bases starting at base `pos` of a WHOLE packed sequence (one padding word after the last one)
__device__ __forceinline__ uint32_t get16(const uint32_t *seq, int pos) {
    const int w = pos >> 4, sh = (pos & 15) << 1;
    const unsigned long long two = ((unsigned long long)seq[w] << 32) | seq[w + 1]; // both words, no branch on sh
    return (uint32_t)((two << sh) >> 32);
}
// Matching bases at (v, h): one step of the greedy extension.  WIN: at most 32, through the sliding windows (both
// positions inside); otherwise at most 16, the whole packed sequences being in LDS (buf = the sequence, w0 = 0).  The
// windows cost 2 KB of LDS whatever the length; whole sequences cost len / 4 bytes but a cheaper step: alignments up to
// 8 kb keep the latter (measured: 128 vs 138 ms per c3-shaped launch), longer ones gain more from the residency.
template <bool WIN> __device__ __forceinline__ int wfa_match_run(const WfaWin &Q, const WfaWin &T, int v, int h) {
    int nm;
    if (WIN) {
        const uint64_t d = wfa_win_get32(Q, v) ^ wfa_win_get32(T, h);
        nm = __clzll((long long)d) >> 1; // 32 when all 32 bases match (__clzll(0) == 64)
    } else {
        const uint32_t d = get16(Q.buf, v) ^ get16(T.buf, h);
        nm = d ? (__clz(d) >> 1) : 16;
    }
    const int rem = Q.len - v < T.len - h ? Q.len - v : T.len - h;
    nm = nm < rem ? nm : rem;
    return nm > 0 ? nm : 0;
}
__device__ __forceinline__ int wave_min_i32_slow(int v) { // rare paths only
    for (int o = 32; o > 0; o >>= 1) {
        const int x = __shfl_xor(v, o, 64);
        v = x < v ? x : v;
    }
    return v;
}

__device__ __forceinline__ unsigned long long rotr64(unsigned long long x, int r) {
    r &= 63;
    return r ? ((x >> r) | (x << (64 - r))) : x;
}

// ---- backtrace of the LDS wavefront kernel ----------------------------------------------------------------------------
// The forward pass stores ONE BYTE per wavefront cell instead of the three 32-bit offsets (M, I, D) of WFA2 / lm_wfa_align:
//   bits 0-1  where M[s][k] came from: 0 mismatch (M[s-4][k]), 1 insertion (I[s][k]), 2 deletion (D[s][k]), with the
//             priority of lm_wfa_backtrace on equal offsets (mismatch > D > I)
//   bit 2     I[s][k] came from I[s-2][k-1] (extension) rather than M[s-8][k-1] (open); ties -> extension
//   bit 3     the same for D[s][k]
// and per even score {first diagonal, byte offset of the row}.  The walk from (final score, final diagonal) to score 0
// follows those codes - it needs no offsets: which cell precedes which is all that is stored - and yields the edit
// operations in reverse; the match runs between them are recovered by replaying the operations forwards with the same
// greedy extension the forward pass used (every M cell was extended maximally, so the replay lands on the same cells).
// 12x less wavefront traffic than the offsets, and the walk runs out of LDS: rows of ~30 scores are fetched with one
// coalesced copy (their bytes are contiguous), so one global round trip serves ~10-15 operations instead of two round
// trips per operation.
#define BT_WIN 4032 /* bytes of backtrace rows held in LDS during the walk (BtLds fits the 128-diagonal ring it reuses) */
template <int BW> struct BtLdsT { // BW: multiple of 16
    uint8_t win[BW + 32];
    int32_t lo[64], base[64];
};
typedef BtLdsT<BT_WIN> BtLds;

// Returns the number of edit operations written (descending from opseq_end), or -1 on overflow / inconsistency.
// ops bytes: bits 0-1 = 0 X, 1 I, 2 D; bit 2 = the cell the operation arrives at is an M cell (extend after it).
template <int BW>
__device__ __forceinline__ int bt_walk(const int32_t *__restrict__ hdr2, const uint8_t *__restrict__ bt, int s_final, int ak,
                                       uint8_t *__restrict__ opseq_end, int64_t opseq_room, BtLdsT<BW> *L, int lane) {
    int score = s_final, k = ak, matrix = 0;
    int64_t nops = 0;
    while (score > 0) {
        // window: rows of scores score, score-2, ... as far down as BW bytes reach (at most 64 rows)
        const int top = score >> 1;
        int32_t lo_j = 0, base_j = 0;
        if (top - lane >= 0) {
            lo_j = hdr2[2 * (top - lane)];
            base_j = hdr2[2 * (top - lane) + 1];
        }
        const int32_t top_end = __builtin_amdgcn_readfirstlane(hdr2[2 * (top + 1) + 1]);
        const unsigned long long fits = __ballot(top - lane >= 0 && top_end - base_j <= BW);
        // rows are contiguous and in score order: the lanes that fit form a prefix
        const unsigned long long nfit = ~fits;
        const int nrows = nfit ? __ffsll((long long)nfit) - 1 : 64;
        if (nrows < 1) return -1;
        const int32_t wbase = __builtin_amdgcn_readfirstlane(__shfl(base_j, nrows - 1));
        LDS_WAVE_SYNC();
        L->lo[lane] = lo_j;
        L->base[lane] = base_j;
        {
            const int32_t a0 = wbase & ~15; // aligned 16-byte copies
            const int nchunks = (top_end - a0 + 15) >> 4;
            for (int c = lane; c < nchunks; c += 64) {
                const uint4 v = *(const uint4 *)(bt + a0 + 16 * c);
                *(uint4 *)(L->win + 16 * c) = v;
            }
            LDS_WAVE_SYNC();
            const int32_t shift = wbase - a0; // win[shift + (byte - wbase)]
            const int low = score - 2 * (nrows - 1);
            while (score >= low && score > 0) {
                const int j = (2 * top - score) >> 1;
                const int32_t rlo = L->lo[j], rbase = L->base[j];
                const int code = L->win[shift + (rbase - wbase) + (k - rlo)];
                int op, ext;
                if (matrix == 0) {
                    op = code & 3;
                    ext = op == 1 ? (code >> 2) & 1 : (code >> 3) & 1;
                } else {
                    op = matrix;
                    ext = matrix == 1 ? (code >> 2) & 1 : (code >> 3) & 1;
                }
                if (op == 3 || k < rlo) return -1;
                if (nops >= opseq_room) return -1;
                nops++;
                if (lane == 0) opseq_end[-nops] = (uint8_t)(op | (matrix == 0 ? 4 : 0));
                if (op == 0) {
                    score -= 4;
                    matrix = 0;
                } else {
                    score -= ext ? 2 : 8;
                    k += op == 1 ? -1 : 1;
                    matrix = ext ? op : 0;
                }
                score = __builtin_amdgcn_readfirstlane(score);
                k = __builtin_amdgcn_readfirstlane(k);
                matrix = __builtin_amdgcn_readfirstlane(matrix);
            }
        }
    }
    if (score != 0 || k != 0 || matrix != 0) return -1;
    return (int)nops;
}

// Forward replay of the edit operations: match runs by greedy extension over the 2-bit packed sequences in LDS, runs merged
// like lm_wfa_backtrace's push, alignment statistics of the M-trimmed run list (lib-index-search.go:2278-2302) and the
// BLAST-style score (lib-index-search-util.go:260-304) accumulated on the way; runs stored only when `ops` is given.
template <bool WIN>
__device__ __forceinline__ void bt_replay(const uint8_t *__restrict__ opseq, int nops, WfaWin &Q, WfaWin &T,
                                          int plen, int tlen, uint64_t *__restrict__ ops, int ops_cap, int lane, int s_final,
                                          LmWfaOut *out, int *blast) {
    bool bad = false;
    int v = 0, h = 0;
    int cur_op = 0, cur_n = 0, run_q = 0, run_t = 0; // pending run and where it starts
    bool seen_m = false, overflow = false;
    int wp = 0;
    int alen = 0, matches = 0, gaps = 0, greg = 0, bl = 0;          // since the first M run
    int c_alen = 0, c_matches = 0, c_gaps = 0, c_greg = 0, c_bl = 0; // up to the last M run
    int qbegin = 0, tbegin = 0, qend = 0, tend = 0;
    auto flush = [&]() {
        if (cur_n == 0) return;
        if (ops) {
            if (wp >= ops_cap)
                overflow = true;
            else if (lane == 0)
                ops[wp] = ((uint64_t)(uint32_t)cur_op << 32) | (uint32_t)cur_n;
            wp++;
        }
        if (cur_op == 'M') {
            if (!seen_m) {
                seen_m = true;
                qbegin = run_q + 1;
                tbegin = run_t + 1;
            }
            alen += cur_n;
            matches += cur_n;
            bl += 2 * cur_n;
            c_alen = alen;
            c_matches = matches;
            c_gaps = gaps;
            c_greg = greg;
            c_bl = bl;
            qend = run_q + cur_n;
            tend = run_t + cur_n;
        } else if (seen_m) {
            alen += cur_n;
            if (cur_op == 'X') {
                bl -= 3 * cur_n;
            } else {
                gaps += cur_n;
                greg++;
                bl -= 5 + 2 * cur_n;
            }
        }
    };
    auto extend = [&]() { // v and h are the same in every lane: the windows simply follow them (forwards; the first call
        int run = 0;      // brings them back from where the forward pass left them)
        while (true) {
            if (plen - v <= 0 || tlen - h <= 0) break;
            if (WIN && !(wfa_win_has(Q, v) && wfa_win_has(T, h))) wfa_win_move2(Q, v >> 4, T, h >> 4, lane, &bad, false);
            const int nm = wfa_match_run<WIN>(Q, T, v, h);
            v += nm;
            h += nm;
            run += nm;
            if (nm < (WIN ? 32 : 16)) break;
        }
        return run;
    };
    {
        const int q0 = v, t0 = h;
        const int r = extend();
        if (r > 0) {
            cur_op = 'M';
            cur_n = r;
            run_q = q0;
            run_t = t0;
        }
    }
    for (int c = 0; c < nops; c += 64) {
        const int mine = c + lane < nops ? (int)opseq[c + lane] : 0;
        const int lim = nops - c < 64 ? nops - c : 64;
        for (int i = 0; i < lim; i++) {
            const int ob = __builtin_amdgcn_readfirstlane(__shfl(mine, i));
            const int op = ob & 3;
            const int q0 = v, t0 = h;
            // the run starts where the operation starts
            if (op == 0) {
                if (cur_op != 'X') {
                    flush();
                    cur_op = 'X';
                    cur_n = 0;
                    run_q = q0;
                    run_t = t0;
                }
                cur_n++;
                v++;
                h++;
            } else if (op == 1) {
                if (cur_op != 'I') {
                    flush();
                    cur_op = 'I';
                    cur_n = 0;
                    run_q = q0;
                    run_t = t0;
                }
                cur_n++;
                h++;
            } else {
                if (cur_op != 'D') {
                    flush();
                    cur_op = 'D';
                    cur_n = 0;
                    run_q = q0;
                    run_t = t0;
                }
                cur_n++;
                v++;
            }
            if (ob & 4) {
                const int q1 = v, t1 = h;
                const int r = extend();
                if (r > 0) {
                    flush();
                    cur_op = 'M';
                    cur_n = r;
                    run_q = q1;
                    run_t = t1;
                }
            }
        }
    }
    flush();
    out->status = 0;
    out->score = s_final;
    out->nops = ops ? wp : 0;
    out->qbegin = qbegin;
    out->tbegin = tbegin;
    out->qend = qend;
    out->tend = tend;
    out->align_len = (uint32_t)c_alen;
    out->matches = (uint32_t)c_matches;
    out->gaps = (uint32_t)c_gaps;
    out->gap_regions = (uint32_t)c_greg;
    *blast = c_bl;
    if (overflow || v != plen || h != tlen) {
        out->status = 1;
        out->nops = 0;
    } else if (!seen_m) {
        out->status = 2;
    }
    if (__ballot(bad) != 0ull) { // a byte that is not A/C/G/T was packed on the way: the byte-comparing kernel takes it
        out->status = 3;
        out->nops = 0;
    }
}

// ---- the LDS wavefront kernels -----------------------------------------------------------------------------------------
// k_wfa_lean2<NC, RT, WIN> (lm_wfa_lean2.h: one wavefront per alignment, 64 * NC diagonals).  Persistent: each wavefront owns a
// private header / arena region and pops problems from a queue ordered by decreasing expected cost; a ring that turns out too
// narrow returns status 3 and the next width takes the problem (... -> k_wfa_wave, the global-memory ring).  Results are
// identical to lm_wfa_align.  (Removed after their A/B: k_wfa_lean / k_wfa_mw - a wrapping ring, five DPP range reductions and
// three LDS hand-offs per score - in round 5; k_wfa_mw2, four wavefronts per long alignment, in round 6: lm_wfa_dev.h.)

#include "lm_wfa_dev.h"

// ------------------------------------------------------------------------------------------------------------
// host-callable launchers
static inline int grid_for(int64_t n, int block, int maxb = 2048 * 8) {
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > maxb) g = maxb;
    return (int)g;
}

#define LM_LAUNCH_1D(kernel, n, stream, ...) \
    hipLaunchKernelGGL(kernel, dim3(grid_for((n), 256)), dim3(256), 0, stream, __VA_ARGS__)

void launch_extract_kmers(hipStream_t st, const uint8_t *qseq, const int64_t *qoff, const int64_t *posoff, int nq, int K,
                          int64_t total_pos, uint64_t *keys_all, uint32_t *vals_all, uint64_t *keys_cmp,
                          uint32_t *vals_cmp, int32_t *nvalid) {
    LM_LAUNCH_1D(k_extract_kmers, total_pos, st, qseq, qoff, posoff, nq, K, keys_all, vals_all, keys_cmp, vals_cmp, nvalid);
}
void launch_fill_u32(hipStream_t st, uint32_t *p, int64_t n, uint32_t v) { LM_LAUNCH_1D(k_fill_u32, n, st, p, n, v); }
void launch_mask(hipStream_t st, const uint64_t *keys_all, const int64_t *posoff, int nq, int M, int K,
                 const uint64_t *masks, uint64_t *out_kmers, int64_t *out_lo, int64_t *out_hi, uint32_t *first_mask) {
    LM_LAUNCH_1D(k_mask, (int64_t)nq * M, st, keys_all, posoff, nq, M, K, masks, out_kmers, out_lo, out_hi, first_mask);
}
void launch_lookup_prep(hipStream_t st, DevIndexView ix, const uint64_t *kmers, const int64_t *klo, const uint32_t *first_mask,
                        int64_t nqm, uint32_t *keys, uint32_t *slots, unsigned long long *counter) {
    const int64_t tiles = (nqm * 2 + 2047) / 2048;
    hipLaunchKernelGGL(k_lookup_prep, dim3((unsigned)std::min<int64_t>(std::max<int64_t>(tiles, 1), 65536)), dim3(256), 0, st, ix,
                       kmers, klo, first_mask, nqm, keys, slots, counter);
}
static int lookup_grid(int64_t total) { // one thread per lookup, workgroup count a multiple of 8 (see lookup_index)
    int64_t nb = (total + 255) / 256;
    nb = (nb + 7) / 8 * 8;
    return (int)(nb < 8 ? 8 : nb);
}
void launch_lookup_count(hipStream_t st, DevIndexView ix, const uint64_t *kmers, const int64_t *klo, const int64_t *khi,
                         const uint32_t *skeys, const uint32_t *sslots, int64_t nlk, int min_prefix, uint32_t *counts,
                         int64_t *starts, int32_t *nscan, unsigned long long *stat_values, uint64_t *lkey, uint64_t *lrec) {
    hipLaunchKernelGGL(k_lookup_count, dim3(lookup_grid(nlk)), dim3(256), 0, st, ix, kmers, klo, khi, skeys, sslots, nlk,
                       min_prefix, counts, starts, nscan, stat_values, lkey, lrec);
}
void launch_lookup_emit_flat(hipStream_t st, DevIndexView ix, const uint32_t *vals_all, const uint32_t *skeys, const uint32_t *sslots,
                             int64_t nlk, const uint32_t *counts, const int64_t *offs, const int64_t *starts, const uint64_t *lkey,
                             const uint64_t *lrec, uint64_t *outA, uint64_t *outB) {
    hipLaunchKernelGGL(k_lookup_emit_flat, dim3(lookup_grid(nlk)), dim3(256), 0, st, ix, vals_all, skeys, sslots, nlk, counts, offs,
                       starts, lkey, lrec, outA, outB);
}
void launch_lookup_emit(hipStream_t st, DevIndexView ix, const uint64_t *kmers, const int64_t *klo, const int64_t *khi,
                        const uint32_t *vals_all, const uint32_t *skeys, const uint32_t *sslots, int64_t nlk,
                        const uint32_t *counts, const int64_t *offs, const int64_t *starts, const int32_t *nscan,
                        uint64_t *outA, uint64_t *outB) {
    hipLaunchKernelGGL(k_lookup_emit, dim3(lookup_grid(nlk)), dim3(256), 0, st, ix, kmers, klo, khi, vals_all, skeys, sslots,
                       nlk, counts, offs, starts, nscan, outA, outB);
}
void launch_chain1(hipStream_t st, const uint64_t *B, const int64_t *seg_off, int nseg, LmChainOpt opt, int K, LmSub *subs,
                   uint8_t *marks, uint64_t *msi, uint64_t *s2i, int8_t *dirs, uint8_t *visited, int32_t *chain_off_pool,
                   int32_t *chain_idx_pool, int32_t *seg_n, float *seg_score, int32_t *seg_nch, int32_t *big_list,
                   unsigned int *big_count) {
    // big_list (nseg entries) / big_count (zeroed by the caller): the pairs left to the wave kernel; null = all by lanes
    hipLaunchKernelGGL(k_chain1, dim3(grid_for(nseg, 64)), dim3(64), 0, st, B, seg_off, nseg, opt, K, subs, marks, msi, s2i,
                       dirs, visited, chain_off_pool, chain_idx_pool, seg_n, seg_score, seg_nch, big_list, big_count);
    if (big_list)
        hipLaunchKernelGGL(k_chain1_wave, dim3(2048), dim3(256), 0, st, B, seg_off, opt, K, subs, marks, msi, s2i, dirs, visited,
                           chain_off_pool, chain_idx_pool, seg_n, seg_score, seg_nch, big_list, big_count);
}
void launch_task_count(hipStream_t st, const float *seg_score, const int32_t *seg_nch, const uint8_t *keep, int nseg,
                       float min_score, int32_t *ntask) {
    LM_LAUNCH_1D(k_task_count, nseg, st, seg_score, seg_nch, keep, nseg, min_score, ntask);
}
void launch_make_tasks(hipStream_t st, DevIndexView ix, const uint64_t *segA, const int64_t *seg_off, int nseg,
                       const LmSub *subs, const int32_t *chain_off_pool, const int32_t *chain_idx_pool,
                       const int32_t *ntask, const int64_t *task_off, const int64_t *qoff, int ext_len,
                       int32_t *order_scratch, Task *tasks) {
    hipLaunchKernelGGL(k_make_tasks, dim3(grid_for(nseg, 64)), dim3(64), 0, st, ix, segA, seg_off, nseg, subs,
                       chain_off_pool, chain_idx_pool, ntask, task_off, qoff, ext_len, order_scratch, tasks);
}
void launch_task_wlen(hipStream_t st, const Task *tasks, int64_t ntasks, int32_t *wlen) {
    LM_LAUNCH_1D(k_task_wlen, ntasks, st, tasks, ntasks, wlen);
}
void launch_task_set_woff(hipStream_t st, Task *tasks, int64_t ntasks, const int64_t *woff) {
    LM_LAUNCH_1D(k_task_set_woff, ntasks, st, tasks, ntasks, woff);
}
void launch_extract_windows(hipStream_t st, DevIndexView ix, const Task *tasks, int64_t ntasks, const int32_t *only,
                            uint8_t *wbuf) {
    int g = (int)(ntasks < 1 ? 1 : (ntasks > 1048576 ? 1048576 : ntasks));
    hipLaunchKernelGGL(k_extract_windows, dim3(g), dim3(256), 0, st, ix, tasks, ntasks, only, wbuf);
}
void launch_extract_windows_at(hipStream_t st, DevIndexView ix, const Task *tasks, const int32_t *idx, const int64_t *dest,
                               int64_t n, uint8_t *wbuf) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_extract_windows_at, dim3((unsigned)std::min<int64_t>(n, 65536)), dim3(256), 0, st, ix, tasks, idx, dest, n,
                       wbuf);
}
void launch_build_cmp_tab(hipStream_t st, const uint64_t *keys_cmp, const int64_t *posoff, const int32_t *nvalid, int nq,
                          int K, const int64_t *tab_off, const int32_t *tab_bits, int64_t tab_words, uint32_t *tab) {
    LM_LAUNCH_1D(k_build_cmp_tab, tab_words, st, keys_cmp, posoff, nvalid, nq, K, tab_off, tab_bits, tab);
}
void launch_sum_i32(hipStream_t st, const int32_t *v, int64_t n, unsigned long long *out) {
    int g = (int)((n + 255) / 256);
    g = g < 1 ? 1 : (g > 2048 ? 2048 : g);
    hipLaunchKernelGGL(k_sum_i32, dim3(g), dim3(256), 0, st, v, n, out);
}
void launch_build_cmp_bits(hipStream_t st, const uint64_t *keys_cmp, const int64_t *posoff, const int32_t *nvalid, int nq,
                           int K, const int64_t *bits_off, const int32_t *bits_log, uint32_t *bits) {
    int g = nq < 1 ? 1 : (nq > 65536 ? 65536 : nq);
    hipLaunchKernelGGL(k_build_cmp_bits, dim3(g), dim3(256), 0, st, keys_cmp, posoff, nvalid, nq, K, bits_off, bits_log, bits);
}
void launch_pa_filter(hipStream_t st, DevIndexView ix, const Task *tasks, int64_t ntasks, const uint8_t *wbuf,
                      const int64_t *posoff, const int32_t *nvalid, const uint32_t *cmp_bits, const int64_t *bits_off,
                      const int32_t *bits_log, int K, int min_prefix, unsigned long long *seg_count, int nseg, int64_t seg_cap,
                      uint64_t *cand, unsigned long long *group_counter, int ncu, int seg_by_group, bool roll) {
    const int64_t ngroups = (ntasks + PA_GROUP - 1) / PA_GROUP;
    int g = (int)(ngroups < 1 ? 1 : (ngroups > ncu ? ncu : ngroups));
    static bool lds_set = false;
    if (!lds_set) {
        (void)hipFuncSetAttribute((const void *)k_pa_filter<true>, hipFuncAttributeMaxDynamicSharedMemorySize, PA_LDS_BYTES);
        (void)hipFuncSetAttribute((const void *)k_pa_filter<false>, hipFuncAttributeMaxDynamicSharedMemorySize, PA_LDS_BYTES);
        lds_set = true;
    }
    // ROLL slices a window into pieces of up to PA_SLICE_ROLL = 2048 positions and takes its rolling branch only for K == 31; any
    // other k would fall into the strided branch, whose one-load-per-lane word cache covers 64 x 32 bases = PA_SLICE positions only
    // (the end of a 2048-position slice came out garbled and candidates were dropped): other k run the strided instantiation
    roll = roll && K == 31;
    hipLaunchKernelGGL(roll ? k_pa_filter<true> : k_pa_filter<false>, dim3(g), dim3(PA_THREADS), PA_LDS_BYTES, st, ix, tasks, ntasks, wbuf,
                       posoff, nvalid, cmp_bits, bits_off, bits_log, K, min_prefix, seg_count, nseg, seg_cap, cand, group_counter,
                       seg_by_group);
}
void launch_pa_search(hipStream_t st, DevIndexView ix, const Task *tasks, const uint8_t *wbuf, const uint64_t *keys_cmp,
                      const uint32_t *vals_cmp, const int64_t *posoff, const int32_t *nvalid, const uint32_t *cmp_tab,
                      const int64_t *tab_off, const int32_t *tab_bits, int K, int min_prefix,
                      const unsigned long long *seg_count, int nseg, int64_t seg_cap, const uint64_t *cand,
                      unsigned long long *count, int64_t cap, uint64_t *outA, uint64_t *outB, int qbits, int tbits,
                      int xcd_map) {
    // the candidate counts are only known on the device: a fixed number of blocks per segment.  xcd_map (segments in query
    // order, k_pa_filter's seg_by_group): as many blocks per range of segments as one XCD holds (32 CUs x 8), so that an XCD
    // works on one or two ranges - a few queries - at a time; otherwise a grid that just fills the chip.
    int bps = xcd_map ? 256 / LM_PA_RANGE_SEGS : (256 * 32 + nseg - 1) / nseg;
    const int64_t need = (seg_cap + 255) / 256;
    if (bps > need) bps = (int)(need < 1 ? 1 : need);
    const int nseg8 = xcd_map ? (nseg / LM_PA_RANGE_SEGS + 7) / 8 * 8 * LM_PA_RANGE_SEGS : nseg;
    static const bool ps_dbg = getenv("LM_DEBUG_PA_SEARCH") != nullptr;
    static unsigned long long *d_dbg = nullptr;
    if (ps_dbg && !d_dbg && hipMalloc((void **)&d_dbg, 8 * sizeof(unsigned long long)) != hipSuccess) d_dbg = nullptr;
    if (ps_dbg && d_dbg) (void)hipMemsetAsync(d_dbg, 0, 8 * sizeof(unsigned long long), st);
    hipLaunchKernelGGL(k_pa_search, dim3(nseg8 * bps), dim3(256), 0, st, ix, tasks, wbuf, keys_cmp, vals_cmp, posoff, nvalid, cmp_tab,
                       tab_off, tab_bits, K, min_prefix, seg_count, seg_cap, bps, cand, count, cap, outA, outB, qbits, tbits, nseg,
                       xcd_map, ps_dbg ? d_dbg : nullptr);
    if (ps_dbg && d_dbg) {
        unsigned long long h[8] = {0};
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h, d_dbg, sizeof h, hipMemcpyDeviceToHost);
        fprintf(stderr, "[lm] k_pa_search: %llu candidates in %llu wavefront passes (%.1f per pass), %llu enumeration passes, %llu anchors (%.2f per enumeration pass), wavefront-ms: search %.1f, enumeration %.1f\n",
                h[0], h[1], h[1] ? (double)h[0] / (double)h[1] : 0.0, h[2], h[3], h[2] ? (double)h[3] / (double)h[2] : 0.0, (double)h[4] / 1e5, (double)h[5] / 1e5);
    }
}
void launch_pa_task_off_sorted(hipStream_t st, const uint64_t *sortedA, int shift, int64_t total, int64_t ntasks,
                               int64_t *pa_off) {
    LM_LAUNCH_1D(k_pa_task_off_sorted, ntasks + 1, st, sortedA, shift, total, ntasks, pa_off);
}
void launch_pa_chain(hipStream_t st, const uint64_t *B, const int64_t *pa_off, int64_t ntasks, int K, LmChain2Opt opt,
                     LmSub *subs, uint8_t *marks, uint64_t *msi, int32_t *stack, LmChain2 *out, int32_t *out_n,
                     int32_t *clr_n, int qbits, int tbits) {
    int g = (int)(ntasks < 1 ? 1 : (ntasks > 262144 ? 262144 : ntasks));
    static const bool pa_dbg = getenv("LM_DEBUG_PA_CHAIN") != nullptr; // phase times of k_pa_chain_wave
    static unsigned long long *d_pa_dbg = nullptr; // (its own small buffer: 16 counters)
    if (pa_dbg && !d_pa_dbg && hipMalloc((void **)&d_pa_dbg, 16 * sizeof(unsigned long long)) != hipSuccess) d_pa_dbg = nullptr;
    unsigned long long *dbg = pa_dbg ? d_pa_dbg : nullptr;
    if (dbg) (void)hipMemsetAsync(dbg, 0, 16 * sizeof(unsigned long long), st);
    hipLaunchKernelGGL(k_pa_chain_wave, dim3(g), dim3(64), 0, st, B, pa_off, ntasks, K, opt, subs, marks, msi, stack, out, out_n,
                       clr_n, qbits, tbits, dbg);
    if (dbg) {
        unsigned long long h[16] = {0};
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h, dbg, sizeof h, hipMemcpyDeviceToHost);
        fprintf(stderr, "[lm] k_pa_chain: %lld windows (%llu with a DP), wavefront-ms: clear+trim %.1f, DP %.1f, backtrack %.1f; "
                        "by anchors left {windows, anchors, wavefront-ms}: 2-8 {%llu, %llu, %.1f} 9-64 {%llu, %llu, %.1f} 65-256 {%llu, %llu, %.1f} 257+ {%llu, %llu, %.1f}\n",
                (long long)ntasks, h[3], (double)h[0] / 1e5, (double)h[1] / 1e5, (double)h[2] / 1e5, h[4], h[5], (double)h[6] / 1e5, h[7], h[8],
                (double)h[9] / 1e5, h[10], h[11], (double)h[12] / 1e5, h[13], h[14], (double)h[15] / 1e5);
    }
}
void launch_gather_chain2(hipStream_t st, const LmChain2 *in, const int64_t *pa_off, const int32_t *out_n,
                          const int64_t *res_off, int64_t ntasks, LmChain2 *out) {
    LM_LAUNCH_1D(k_gather_chain2, ntasks, st, in, pa_off, out_n, res_off, ntasks, out);
}
void launch_extend_count(hipStream_t st, const HspIn *hsps, int64_t n, const uint8_t *qseq, const int64_t *qoff,
                         const uint8_t *wbuf, int32_t *cap) {
    hipLaunchKernelGGL(k_extend_count, dim3(grid_for(n, 64)), dim3(64), 0, st, hsps, n, qseq, qoff, wbuf, cap);
}
void launch_extend_wave_cap(hipStream_t st, const int32_t *cap, int64_t n, int32_t *wcap, int64_t nw) {
    hipLaunchKernelGGL(k_extend_wave_cap, dim3(grid_for(nw, 256)), dim3(256), 0, st, cap, n, wcap, nw);
}
int extend_grid_blocks(int64_t n) { // workgroups of k_extend = slabs of the rows / rstart pools
    int64_t nw = (2 * n + 63) / 64;
    return (int)(nw < 1 ? 1 : (nw > 8192 ? 8192 : nw));
}
void launch_extend(hipStream_t st, const HspIn *hsps, int64_t n, const uint8_t *qseq, const int64_t *qoff,
                   const uint8_t *wbuf, const int32_t *cap, const int64_t *woff, uint16_t *subs, int32_t *msi,
                   void *rows_pool, uint32_t *rstart_pool, HspExt *out) {
    // one wavefront per 32 HSPs (64 flanks); subs / msi hold 64 * woff[nw] entries (transposed per wavefront);
    // rows_pool: extend_grid_blocks(n) * LM_EXT_ROWS * 64 * 16 bytes, rstart_pool: the same count of uint32
    hipLaunchKernelGGL(k_extend, dim3(extend_grid_blocks(n)), dim3(64), 0, st, hsps, n, qseq, qoff, wbuf, cap, woff, subs, msi,
                       (LmM128 *)rows_pool, rstart_pool, out);
    hipLaunchKernelGGL(k_extend_fin, dim3(grid_for(n, 256)), dim3(256), 0, st, hsps, n, out);
}
// the kernel instantiations: ring width x (whole sequences in LDS | sliding windows)
typedef void (*WfaLeanFn)(const WfaIn *, int64_t, const int32_t *, int64_t, int32_t *, int64_t, uint8_t *, int64_t, uint64_t *,
                          unsigned int *, int, int, WfaOut *, unsigned long long *);
// r16: 16-bit ring cells (whole-sequence kernels of 128 / 256 diagonals, sequences up to 12 000 bases: lm_kernels.h)
static WfaLeanFn wfa_lean_fn(int nc, bool win, bool r16) { // (lm_wfa_lean2.h)
    // (flavour margins 2 / 4 / 8 / 12 of the dominant instantiation - the free slots either side when the live rows are centred
    // on the chunks of a flavour - measured on one resident c3mini index: 321 / 318 / 315 / 321 ms per step; 8 kept)
    if (r16 && !win && nc == 2) return k_wfa_lean2<2, int16_t, false, L2_SHRINK_MARGIN, 8>; // (64 VGPRs, 8 wavefronts per SIMD: no difference at C3, kept with k_pa_chain_wave's)
    if (r16 && !win && nc == 4) return k_wfa_lean2<4, int16_t, false>;
    switch (nc) {
    case 16: return win ? k_wfa_lean2<16, int32_t, true> : k_wfa_lean2<16, int32_t, false>;
    case 8: return win ? k_wfa_lean2<8, int32_t, true> : k_wfa_lean2<8, int32_t, false>;
    case 4: return win ? k_wfa_lean2<4, int32_t, true> : k_wfa_lean2<4, int32_t, false>;
    case 1: return win ? k_wfa_lean2<1, int32_t, true> : k_wfa_lean2<1, int32_t, false>;
    default: return win ? k_wfa_lean2<2, int32_t, true> : k_wfa_lean2<2, int32_t, false>;
    }
}
bool wfa_r16_ok(int seq_words, int nc, bool win) { return !win && (nc == 2 || nc == 4) && seq_words <= 750; }

static size_t wfa_dyn_lds(int seq_words, bool win) { // two packed sequences with one padding word each (+2: the predicated
    return win ? 0 : (size_t)(2 * (seq_words + 2) + 1) * sizeof(uint32_t); // extension may read one word past; k_wfa_lean2: one word in front)
}
int wfa_resident_blocks(int device, int seq_words, int nc, bool win, bool r16) {
    int nb = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)wfa_lean_fn(nc, win, r16), 64, wfa_dyn_lds(seq_words, win)) != hipSuccess || nb < 1)
        nb = 8;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 1) cus = 256;
    return nb * cus;
}
void launch_wfa(hipStream_t st, const WfaIn *in, int64_t n, const int32_t *todo, int64_t ntodo, int nblocks,
                int32_t *hdr_pool, int64_t hdr_stride, uint8_t *arena_pool, int64_t arena_stride, uint64_t *ops_pool,
                unsigned int *queue, int seq_words, int want_ops, WfaOut *out, int nc, bool win, bool r16, unsigned long long *dbg) {
    hipLaunchKernelGGL(wfa_lean_fn(nc, win, r16), dim3(nblocks), dim3(64), wfa_dyn_lds(seq_words, win), st, in, n, todo, ntodo, hdr_pool,
                       hdr_stride, arena_pool, arena_stride, ops_pool, queue, seq_words, want_ops, out, dbg);
}
void launch_wfa_wide(hipStream_t st, const WfaIn *in, int64_t n, const int32_t *todo, int64_t ntodo, int32_t *hdr_pool,
                     int32_t *arena_pool, uint64_t *ops_pool, WfaOut *out) {
    int g = (int)(ntodo < 1 ? 1 : (ntodo > 65536 ? 65536 : ntodo));
    hipLaunchKernelGGL(k_wfa_wave, dim3(g), dim3(64), 0, st, in, n, todo, ntodo, hdr_pool, arena_pool, ops_pool, out);
}

} // namespace lm
