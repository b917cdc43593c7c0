// lm_builder.hip — synthetic genome set + seed index generated directly in HBM (bench / large-scale test input).
//
// Index BUILDING is outside the hot-path scope; this exists because the benchmark configurations
// (10k x 5 Mb genomes and up) cannot be built by any CPU tool on a fresh box within minutes, and nothing persists on
// the GPU box.  What it produces has the same structure as a reference-built index (lib-index-build.go):
//   * genomes: procedural i.i.d. ACGT ancestors per family; members are substituted (rate U(0,max_div)) and
//     indel-shifted copies; single contig; stored 2-bit MSB-first like genome/genome.go:1471-1508
//   * normal seeds: EXACT LexicHash capture per genome — for every mask the argmin of mask^kmer over both strands, all
//     occurrences, low-complexity captures dropped (lib-index-build.go:1028-1046)
//   * seed-desert filling as the reference does it (lib-index-build.go:1094-1407): every gap >= max_desert between
//     neighbouring seeds is filled every seed_dist bases with the nearest non-low-complexity k-mer (scan 25 up-, then 24
//     downstream, + strand before - strand) that is the capture of a mask when the window [pre - 1000, pos + 1000 + k)
//     alone is masked, stored under the last mask capturing it (tests/test_gpu_builder.py: every mask's list equals what
//     the oracle's writer stores for the same genomes)
//   * reversed (suffix) seeds for every normal and desert seed (lib-index-build.go:776-890)
//   * seed values batch:17|genome:17|pos:28|strand:1|reversed:1, per-mask arrays sorted by k-mer
// The search kernels and the parity tests never depend on this file: parity uses indexes written in the reference's
// on-disk format by the oracle's writer.
#include <cmath>
#include <cstring>

#include "lm_internal.h"
#include "lm_prims.h"

namespace lm {

static inline void bsync(lm_index *ix) {
    hipError_t e = hipStreamSynchronize(ix->st);
    if (e != hipSuccess) throw HipError(std::string("builder sync: ") + hipGetErrorString(e));
}

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ uint64_t hash3(uint64_t a, uint64_t b, uint64_t c) {
    return mix64(mix64(mix64(a) ^ b) ^ c);
}

struct SynthDev {
    uint64_t seed;
    int64_t genomes;  // whole set
    int32_t genome_len, families;
    double max_div;
    int32_t shard_rank, shard_count;
    int64_t nlocal;
    int32_t nblk;     // 512-base blocks per genome
    int64_t gbytes;   // padded bytes per genome
};

__device__ __forceinline__ int64_t global_genome(const SynthDev &sp, int64_t local) {
    return sp.shard_count > 1 ? local * sp.shard_count + sp.shard_rank : local;
}
__device__ __forceinline__ double genome_div(const SynthDev &sp, int64_t g) {
    if (g < sp.families) return 0.0;
    return sp.max_div * ((double)(hash3(sp.seed, 0xD1Full, (uint64_t)g) >> 11) * (1.0 / 9007199254740992.0));
}

// cumulative indel shift per 512-base block
__global__ void k_synth_shifts(SynthDev sp, int16_t *__restrict__ shifts) {
    for (int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; l < sp.nlocal; l += (int64_t)gridDim.x * blockDim.x) {
        int64_t g = global_genome(sp, l);
        double d = genome_div(sp, g);
        // indel events at a tenth of the substitution rate: 8 trials per block
        uint32_t thr = (uint32_t)(fmin(1.0, d * 0.1 * 512.0 / 8.0) * 4294967295.0);
        int sh = 0;
        for (int b = 0; b < sp.nblk; b++) {
            if (g >= sp.families) {
                uint64_t h = hash3(sp.seed, 0x5117ull + (uint64_t)g, (uint64_t)b);
                for (int t = 0; t < 8; t++) {
                    uint64_t hh = mix64(h + t);
                    if ((uint32_t)hh < thr) sh += (hh >> 63) ? 1 : -1;
                }
                if (sh > 30000) sh = 30000;
                if (sh < -30000) sh = -30000;
            }
            shifts[l * sp.nblk + b] = (int16_t)sh;
        }
    }
}

__device__ __forceinline__ uint32_t synth_base(const SynthDev &sp, int64_t g, int64_t i, int sh, uint32_t sub_thr) {
    uint64_t f = (uint64_t)(g % sp.families);
    uint32_t anc = (uint32_t)(hash3(sp.seed, 0xA11Cull + f, (uint64_t)(i + sh)) >> 17) & 3u;
    if (g >= sp.families) {
        uint64_t h = hash3(sp.seed, 0x5B5ull + (uint64_t)g, (uint64_t)i);
        if ((uint32_t)h < sub_thr) anc = (anc + 1u + (uint32_t)((h >> 40) % 3u)) & 3u;
    }
    return anc;
}

// one lane per packed byte (4 bases, first base in bits 7-6)
__global__ void k_synth_genomes(SynthDev sp, const int16_t *__restrict__ shifts, uint8_t *__restrict__ gbits) {
    int64_t nb = ((int64_t)sp.genome_len + 3) >> 2;
    int64_t total = sp.nlocal * nb;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        int64_t l = t / nb, by = t % nb;
        int64_t g = global_genome(sp, l);
        uint32_t thr = (uint32_t)(genome_div(sp, g) * 4294967295.0);
        uint32_t v = 0;
        for (int j = 0; j < 4; j++) {
            int64_t i = by * 4 + j;
            uint32_t b = 0;
            if (i < sp.genome_len) b = synth_base(sp, g, i, shifts[l * sp.nblk + (i >> 9)], thr);
            v = (v << 2) | b;
        }
        gbits[l * sp.gbytes + by] = (uint8_t)v;
    }
}

// 31-mer at base position pos of a packed genome: two ALIGNED 64-bit loads + funnel shift (genomes start on 8-byte
// boundaries and are padded by 16 bytes; an unaligned 8-byte memcpy compiles to byte loads)
__device__ __forceinline__ uint64_t packed_kmer(const uint8_t *gb, int64_t pos, int K) {
    const int64_t byte = pos >> 2;
    const uint64_t *p = (const uint64_t *)(gb + (byte & ~7ll));
    const uint64_t H = __builtin_bswap64(p[0]), L = __builtin_bswap64(p[1]);
    const int o = (int)(byte & 7) * 8 + (int)(pos & 3) * 2; // 0..62
    const uint64_t v = o ? ((H << o) | (L >> (64 - o))) : H;
    return v >> (64 - (K << 1));
}

struct MaskTab {
    const uint64_t *masks;
    const int32_t *pfx_first;
    int K, p, M;
};

// pass A: per genome of the chunk, argmin hash per mask
__global__ void k_cap_argmin(SynthDev sp, MaskTab mt, const uint8_t *__restrict__ gbits, int64_t l0, int nchunk,
                             unsigned long long *__restrict__ hashes) {
    int64_t npos = (int64_t)sp.genome_len - mt.K + 1;
    int64_t total = (int64_t)nchunk * npos;
    int shift = (mt.K - mt.p) << 1;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        int c = (int)(t / npos);
        int64_t pos = t % npos;
        const uint8_t *gb = gbits + (l0 + c) * sp.gbytes;
        uint64_t fwd = packed_kmer(gb, pos, mt.K);
        uint64_t rc = lm_revcomp(fwd, mt.K);
        unsigned long long *hs = hashes + (int64_t)c * mt.M;
        for (int s = 0; s < 2; s++) {
            uint64_t x = s ? rc : fwd;
            uint64_t pf = x >> shift;
            for (int j = mt.pfx_first[pf]; j < mt.pfx_first[pf + 1]; j++) {
                unsigned long long h = mt.masks[j] ^ x;
                if (h < hs[j]) atomicMin(&hs[j], h);
            }
        }
    }
}

// pass B: emit every occurrence of every captured k-mer
__global__ void k_cap_emit(SynthDev sp, MaskTab mt, const uint8_t *__restrict__ gbits, int64_t l0, int nchunk,
                           const unsigned long long *__restrict__ hashes, uint16_t *__restrict__ s_mask,
                           uint64_t *__restrict__ s_kmer, uint64_t *__restrict__ s_val, unsigned long long *__restrict__ counter,
                           unsigned long long cap, uint64_t *__restrict__ pos_keys, unsigned long long *__restrict__ pos_counter,
                           unsigned long long pos_cap) {
    int64_t npos = (int64_t)sp.genome_len - mt.K + 1;
    int64_t total = (int64_t)nchunk * npos;
    int shift = (mt.K - mt.p) << 1;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        int c = (int)(t / npos);
        int64_t pos = t % npos;
        const uint8_t *gb = gbits + (l0 + c) * sp.gbytes;
        uint64_t fwd = packed_kmer(gb, pos, mt.K);
        uint64_t rc = lm_revcomp(fwd, mt.K);
        const unsigned long long *hs = hashes + (int64_t)c * mt.M;
        int64_t g = global_genome(sp, l0 + c);
        uint64_t bg = ((uint64_t)(g / 5000) << 17) | (uint64_t)(g % 5000);
        for (int s = 0; s < 2; s++) {
            uint64_t x = s ? rc : fwd;
            uint64_t pf = x >> shift;
            bool lc_known = false, lc = false;
            for (int j = mt.pfx_first[pf]; j < mt.pfx_first[pf + 1]; j++) {
                if ((mt.masks[j] ^ x) != hs[j]) continue;
                if (!lc_known) {
                    lc = x == 0 || lm_low_complexity(x, mt.K);
                    lc_known = true;
                }
                if (lc) continue;
                unsigned long long o = atomicAdd(counter, 1ull);
                if (o < cap) {
                    s_mask[o] = (uint16_t)j;
                    s_kmer[o] = x;
                    s_val[o] = (bg << 30) | ((uint64_t)pos << 2) | ((uint64_t)s << 1);
                }
                unsigned long long po = atomicAdd(pos_counter, 1ull);
                if (po < pos_cap) pos_keys[po] = ((uint64_t)c << 32) | ((uint64_t)pos << 1) | (uint64_t)s;
            }
        }
    }
}

// LexicHash capture of one genome by one workgroup, both passes in one launch: the per-mask minima live in LDS
// (M x 8 B = 160 KB for the default 20000 masks: the whole LDS of a CU) instead of a global table hammered with atomics.
// Masks of a p-base prefix are found without a table in memory: in a lexicmap mask set every prefix has one mask and
// some have two (docs/content/usage/utils/masks.md:69-110), so first(pf) = pf + #doubled prefixes below pf: a 4^p-bit
// map + per-word counts (2.5 KB).  Phase 1: ds_min_u64 of mask^kmer; phase 2: every k-mer equal to its mask's minimum is
// emitted (all occurrences, lib-index-build.go:1028-1046), low-complexity captures dropped.
__global__ __launch_bounds__(1024) void k_capture_lds(SynthDev sp, MaskTab mt, const uint8_t *__restrict__ gbits, int64_t l0,
                                                      const uint64_t *__restrict__ dbl_map, const uint32_t *__restrict__ dbl_cnt,
                                                      uint16_t *__restrict__ s_mask, uint64_t *__restrict__ s_kmer,
                                                      uint64_t *__restrict__ s_val, unsigned long long *__restrict__ counter,
                                                      unsigned long long cap, uint64_t *__restrict__ pos_keys,
                                                      unsigned long long *__restrict__ pos_counter, unsigned long long pos_cap) {
    extern __shared__ unsigned long long lds_dyn[];
    unsigned long long *hs = lds_dyn;                                   // [M]
    const int nw = ((1 << (2 * mt.p)) + 63) >> 6;
    unsigned long long *bm = hs + mt.M;                                  // [nw]
    uint32_t *bc = (uint32_t *)(bm + nw);                                // [nw]
    const int c = blockIdx.x;
    const uint8_t *gb = gbits + (l0 + c) * sp.gbytes;
    const int64_t npos = (int64_t)sp.genome_len - mt.K + 1;
    const int shift = (mt.K - mt.p) << 1;
    for (int i = threadIdx.x; i < mt.M; i += blockDim.x) hs[i] = ~0ull;
    for (int i = threadIdx.x; i < nw; i += blockDim.x) {
        bm[i] = dbl_map[i];
        bc[i] = dbl_cnt[i];
    }
    __syncthreads();
    for (int64_t pos = threadIdx.x; pos < npos; pos += blockDim.x) {
        const uint64_t fwd = packed_kmer(gb, pos, mt.K);
        const uint64_t rc = lm_revcomp(fwd, mt.K);
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const uint64_t x = s ? rc : fwd;
            const uint32_t pf = (uint32_t)(x >> shift);
            const unsigned long long w = bm[pf >> 6];
            const int j0 = (int)(pf + bc[pf >> 6] + (uint32_t)__popcll(w & ((1ull << (pf & 63)) - 1)));
            const int nj = 1 + (int)((w >> (pf & 63)) & 1ull);
            for (int j = j0; j < j0 + nj; j++) {
                const unsigned long long h = mt.masks[j] ^ x;
                if (h < hs[j]) atomicMin(&hs[j], h);
            }
        }
    }
    __syncthreads();
    const int64_t g = global_genome(sp, l0 + c);
    const uint64_t bg = ((uint64_t)(g / 5000) << 17) | (uint64_t)(g % 5000);
    // phase 2a counts this thread's captures, the workgroup reserves ONE contiguous range of the staging arrays for the
    // genome (a per-capture atomic on the shared counter costs more than the whole sweep), phase 2b writes
    uint32_t *wsum = bc + nw; // [16] wave totals
    __shared__ unsigned long long base_seed, base_pos;
    uint32_t mine = 0;
    for (int sweep = 0; sweep < 2; sweep++) {
        unsigned long long o = 0, po = 0;
        if (sweep == 1) {
            uint32_t incl = mine;
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t v = __shfl_up(incl, d);
                if (lane >= d) incl += v;
            }
            if (lane == 63) wsum[wave] = incl;
            __syncthreads();
            uint32_t before = 0, all = 0;
            for (int w2 = 0; w2 < (int)(blockDim.x >> 6); w2++) {
                if (w2 < wave) before += wsum[w2];
                all += wsum[w2];
            }
            if (threadIdx.x == 0) {
                base_seed = atomicAdd(counter, (unsigned long long)all);
                base_pos = atomicAdd(pos_counter, (unsigned long long)all);
            }
            __syncthreads();
            o = base_seed + before + (incl - mine);
            po = base_pos + before + (incl - mine);
        }
        for (int64_t pos = threadIdx.x; pos < npos; pos += blockDim.x) {
            const uint64_t fwd = packed_kmer(gb, pos, mt.K);
            const uint64_t rc = lm_revcomp(fwd, mt.K);
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const uint64_t x = s ? rc : fwd;
                const uint32_t pf = (uint32_t)(x >> shift);
                const unsigned long long w = bm[pf >> 6];
                const int j0 = (int)(pf + bc[pf >> 6] + (uint32_t)__popcll(w & ((1ull << (pf & 63)) - 1)));
                const int nj = 1 + (int)((w >> (pf & 63)) & 1ull);
                for (int j = j0; j < j0 + nj; j++) {
                    if ((mt.masks[j] ^ x) != hs[j]) continue;
                    if (x == 0 || lm_low_complexity(x, mt.K)) continue;
                    if (sweep == 0) {
                        mine++;
                        continue;
                    }
                    if (o < cap) {
                        s_mask[o] = (uint16_t)j;
                        s_kmer[o] = x;
                        s_val[o] = (bg << 30) | ((uint64_t)pos << 2) | ((uint64_t)s << 1);
                    }
                    if (po < pos_cap) pos_keys[po] = ((uint64_t)c << 32) | ((uint64_t)pos << 1) | (uint64_t)s;
                    o++;
                    po++;
                }
            }
        }
    }
}

__global__ void k_pseudo_pos(int nchunk, int32_t last_pos, uint64_t *__restrict__ pos_keys, unsigned long long base) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < nchunk) pos_keys[base + c] = ((uint64_t)c << 32) | ((uint64_t)(uint32_t)last_pos << 1) | 1ull; // sorts last
}

__device__ __forceinline__ int closest_mask(const MaskTab &mt, uint64_t x) {
    uint64_t pf = x >> ((mt.K - mt.p) << 1);
    int minj = -1;
    uint64_t minh = ~0ull;
    for (int j = mt.pfx_first[pf]; j < mt.pfx_first[pf + 1]; j++) {
        uint64_t h = mt.masks[j] ^ x;
        if (h < minh) {
            minh = h;
            minj = j;
        }
    }
    return minj;
}

// Desert filling, lib-index-build.go:1094-1407, as the reference does it: for every pair of neighbouring seeds at least
// max_desert apart, walk from pre + seed_dist in steps of seed_dist; at each step scan seed_pos_r positions upstream, then
// downstream, for a non-low-complexity k-mer (+ strand before - strand) that IS THE CAPTURE OF SOME MASK WHEN THE WINDOW
// [pre - 1000, pos + 1000 + k) ALONE IS MASKED (MaskKnownDistinctPrefixes(window, nil, false), :1191-1240), and store it
// under that mask - the LAST (largest-index) mask that captures it.  A wavefront takes 64 seed pairs, finds the deserts
// among them and walks them one after the other; the capture test of a candidate is a sweep of the window by the 64 lanes
// (is any window k-mer of either strand with the same p-base prefix closer to the mask?).
// the p-base prefixes (p <= 16) of the k-mer at `pos` and of its reverse complement, from one 32-base window of the 2-bit
// genome: all the sweep below needs for the ~16 000 : 1 window k-mers that do not share the candidate's prefix
__device__ __forceinline__ void kmer_prefixes(const uint8_t *gb, int64_t pos, int K, int p, uint32_t *fwd, uint32_t *rc) {
    const int64_t byte = pos >> 2;
    const uint64_t *q = (const uint64_t *)(gb + (byte & ~7ll));
    const uint64_t H = __builtin_bswap64(q[0]), L = __builtin_bswap64(q[1]);
    const int o = (int)(byte & 7) * 8 + (int)(pos & 3) * 2; // 0..62
    const uint64_t v = o ? ((H << o) | (L >> (64 - o))) : H; // 32 bases from pos, left aligned
    *fwd = (uint32_t)(v >> (64 - 2 * p));
    const uint32_t last = (uint32_t)(v >> (64 - 2 * K)) & ((1u << (2 * p)) - 1u); // the last p bases of the k-mer
    uint32_t y = __builtin_bitreverse32(~last);                                    // complement, reverse the bits ...
    y = ((y >> 1) & 0x55555555u) | ((y & 0x55555555u) << 1);                       // ... and put the base pairs back in order
    *rc = y >> (32 - 2 * p);
}
__device__ __forceinline__ int desert_capturing_mask(const MaskTab &mt, const uint8_t *gb, int64_t wstart, int nk, uint64_t x,
                                                     int lane) {
    const int shift = (mt.K - mt.p) << 1;
    const uint32_t pf = (uint32_t)(x >> shift);
    const int j0 = mt.pfx_first[pf], j1 = mt.pfx_first[pf + 1];
    if (j0 >= j1) return -1;
    // one sweep of the window for all (one or two) masks of the prefix: bit j - j0 = x is beaten for mask j
    uint32_t beaten = 0;
    for (int w = lane; w < nk; w += 64) {
        uint32_t pfw, prc;
        kmer_prefixes(gb, wstart + w, mt.K, mt.p, &pfw, &prc);
        if (pfw == pf || prc == pf) {
            const uint64_t f = packed_kmer(gb, wstart + w, mt.K), r = lm_revcomp(f, mt.K);
            for (int j = j0; j < j1 && j - j0 < 32; j++) {
                const uint64_t mk = mt.masks[j], hx = mk ^ x;
                if ((pfw == pf && (mk ^ f) < hx) || (prc == pf && (mk ^ r) < hx)) beaten |= 1u << (j - j0);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) beaten |= (uint32_t)__shfl_xor((int)beaten, o, 64);
    int im = -1;
    for (int j = j0; j < j1 && j - j0 < 32; j++)
        if (!((beaten >> (j - j0)) & 1u)) im = j; // x attains the window's minimum for mask j; the last such mask is recorded
    return im;
}
__global__ __launch_bounds__(256) void k_desert_fill(SynthDev sp, MaskTab mt, const uint8_t *__restrict__ gbits, int64_t l0,
                                                      const uint64_t *__restrict__ pos_keys, int64_t npk, int max_desert,
                                                      int seed_dist, uint16_t *__restrict__ s_mask,
                                                      uint64_t *__restrict__ s_kmer, uint64_t *__restrict__ s_val,
                                                      unsigned long long *__restrict__ counter, unsigned long long cap) {
    const int seed_pos_r = seed_dist / 2;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t base = wave * 64; base < npk; base += nwaves * 64) {
        const int64_t t = base + lane;
        int c = 0, pos = 0, pre = 0;
        bool isd = false;
        if (t < npk) {
            const uint64_t key = pos_keys[t];
            c = (int)(key >> 32);
            pos = (int)((key & 0xffffffffu) >> 1);
            if (t > 0 && (int)(pos_keys[t - 1] >> 32) == c) pre = (int)((pos_keys[t - 1] & 0xffffffffu) >> 1);
            isd = pos - pre >= max_desert;
        }
        uint64_t todo = __ballot(isd);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int gc = __builtin_amdgcn_readfirstlane(__shfl(c, src, 64));
            const int gpos = __builtin_amdgcn_readfirstlane(__shfl(pos, src, 64));
            const int gpre = __builtin_amdgcn_readfirstlane(__shfl(pre, src, 64));
            const uint8_t *gb = gbits + (l0 + gc) * sp.gbytes;
            const int64_t g = global_genome(sp, l0 + gc);
            const uint64_t bg = ((uint64_t)(g / 5000) << 17) | (uint64_t)(g % 5000);
            // the window that is masked on its own (:1150-1190)
            int wstart = gpre - 1000;
            if (wstart < 0) wstart = 0;
            int wend = gpos + 1000 + mt.K;
            if (wend > sp.genome_len) wend = sp.genome_len;
            const int nk = wend - wstart - mt.K + 1; // k-mers of the window
            auto try_at = [&](int at, uint64_t *kmer, int *strand, int *im) { // the candidate at genome position `at`
                const int rel = at - wstart;
                if (rel < 0 || rel >= nk) return false;
                const uint64_t f = packed_kmer(gb, at, mt.K);
                if (f != 0 && !lm_low_complexity(f, mt.K)) {
                    const int m = desert_capturing_mask(mt, gb, wstart, nk, f, lane);
                    if (m >= 0) {
                        *kmer = f;
                        *strand = 0;
                        *im = m;
                        return true;
                    }
                }
                const uint64_t r = lm_revcomp(f, mt.K);
                if (r != 0 && !lm_low_complexity(r, mt.K)) {
                    const int m = desert_capturing_mask(mt, gb, wstart, nk, r, lane);
                    if (m >= 0) {
                        *kmer = r;
                        *strand = 1;
                        *im = m;
                        return true;
                    }
                }
                return false;
            };
            int j = gpre + seed_dist;
            while (j < gpos) {
                const int start_dn = j + 1, end_up = j - seed_pos_r;
                bool ok = false;
                uint64_t kmer = 0;
                int strand = 0, im = -1, at = j;
                for (; at > end_up; at--)
                    if (try_at(at, &kmer, &strand, &im)) {
                        ok = true;
                        break;
                    }
                if (!ok) {
                    if (start_dn >= gpos) break;
                    int end_dn = start_dn + seed_pos_r;
                    if (end_dn >= gpos) end_dn = gpos - 1;
                    for (at = start_dn; at < end_dn; at++)
                        if (try_at(at, &kmer, &strand, &im)) {
                            ok = true;
                            break;
                        }
                }
                if (ok && lane == 0) {
                    const unsigned long long o = atomicAdd(counter, 1ull);
                    if (o < cap) {
                        s_mask[o] = (uint16_t)im;
                        s_kmer[o] = kmer;
                        s_val[o] = (bg << 30) | ((uint64_t)at << 2) | ((uint64_t)strand << 1);
                    }
                }
                j = at + seed_dist;
            }
        }
    }
}

// reversed copies of seeds [from, to)
__global__ void k_reverse_seeds(MaskTab mt, unsigned long long from, unsigned long long to, uint16_t *__restrict__ s_mask,
                                uint64_t *__restrict__ s_kmer, uint64_t *__restrict__ s_val,
                                unsigned long long *__restrict__ counter, unsigned long long cap) {
    for (unsigned long long t = from + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; t < to;
         t += (unsigned long long)gridDim.x * blockDim.x) {
        uint64_t rev = lm_reverse(s_kmer[t], mt.K);
        int m = closest_mask(mt, rev);
        unsigned long long o = atomicAdd(counter, 1ull);
        if (o < cap && m >= 0) {
            s_mask[o] = (uint16_t)m;
            s_kmer[o] = rev;
            s_val[o] = s_val[t] | 1ull;
        }
    }
}

__global__ void k_fill_u64(unsigned long long *p, int64_t n, unsigned long long v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

__global__ void k_fetch_bases(const uint8_t *__restrict__ gb, int64_t start, int64_t len, uint8_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t p = start + i;
        out[i] = (uint8_t)("ACGT"[(gb[p >> 2] >> ((3 - (p & 3)) << 1)) & 3]);
    }
}

static int gridn(int64_t n, int block = 256) {
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > 262144) g = 262144;
    return (int)g;
}

// host-side mask set with the structure of lexicmap masks (docs/content/usage/utils/masks.md:69-110): all p-prefixes
// present, M-4^p extra masks on distinct prefixes differing from their twin at base p+1, sorted.
static void gen_masks(int k, int M, uint64_t seed, std::vector<uint64_t> &out) {
    int p = std::max(1, (int)(std::log2((double)M) / 2));
    int64_t np = (int64_t)1 << (2 * p);
    int lowbits = (k - p) << 1;
    uint64_t lowmask = (1ull << lowbits) - 1;
    uint64_t st = seed * 0x2545F4914F6CDD1Dull + 0x9E37ull;
    auto next = [&]() {
        st += 0x9E3779B97F4A7C15ull;
        return mix64(st);
    };
    out.clear();
    for (int64_t i = 0; i < np && (int)out.size() < M; i++) {
        uint64_t m;
        do m = ((uint64_t)i << lowbits) | (next() & lowmask);
        while (lm_dust(m, k));
        out.push_back(m);
    }
    int extra = M - (int)out.size();
    std::vector<int64_t> perm(np);
    for (int64_t i = 0; i < np; i++) perm[i] = i;
    for (int j = 0; j < extra; j++) {
        int64_t r = j + (int64_t)(next() % (uint64_t)(np - j));
        std::swap(perm[j], perm[r]);
        uint64_t twin_base = (out[perm[j]] >> (lowbits - 2)) & 3;
        uint64_t m;
        do m = ((uint64_t)perm[j] << lowbits) | (next() & lowmask);
        while (((m >> (lowbits - 2)) & 3) == twin_base || lm_dust(m, k));
        out.push_back(m);
    }
    std::sort(out.begin(), out.end());
}

} // namespace lm

extern "C" {

lm_status lm_index_build_synthetic(const lm_synth_spec *spec, const lm_options *opt, int device, lm_index **out) {
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        g_open_error = "no HIP device available (this library has no CPU path)";
        return LM_ERR_NO_DEVICE;
    }
    if (spec->k != 31 || spec->masks < 4 || spec->masks > 65535 || spec->genome_len < 64 || spec->genomes < 1 ||
        spec->genome_len >= (1 << 28) || spec->families < 1) {
        g_open_error = "lm_index_build_synthetic: unsupported spec (k must be 31, masks in [4,65535])";
        return LM_ERR_ARG;
    }
    lm_index *ix = new lm_index();
    try {
        ix->opt = *opt;
        ix->device = device;
        HIPCHK(hipSetDevice(device));
        HIPCHK(hipStreamCreate(&ix->st));
        HostIndex &h = ix->host;
        const int K = spec->k, M = spec->masks;
        h.k = K;
        h.M = M;
        h.main_version = 3;
        h.minor_version = 5;
        h.synthetic = true;
        h.synth_genome_len = spec->genome_len;
        h.synth_genomes = spec->genomes;
        h.mask_prefix = std::max(1, (int)(std::log2((double)M) / 2));
        h.anchor_prefix = 6;
        h.contig_interval = 1000;
        h.shard_rank = opt->shard_count > 1 ? opt->shard_rank : 0;
        h.shard_count = opt->shard_count > 1 ? opt->shard_count : 1;
        h.total_bases = opt->total_bases_override > 0 ? opt->total_bases_override : spec->genomes * (int64_t)spec->genome_len;
        gen_masks(K, M, (uint64_t)spec->mask_seed, h.masks);
        const int p = h.mask_prefix;
        std::vector<int32_t> pfx((size_t)(1ull << (2 * p)) + 1, 0);
        for (int i = 0; i < M; i++) pfx[(size_t)(h.masks[i] >> ((K - p) << 1)) + 1]++;
        for (size_t i = 1; i < pfx.size(); i++) pfx[i] += pfx[i - 1];
        {
            if (opt->min_prefix > K || opt->min_prefix < p + h.anchor_prefix) {
                g_open_error = "MinPrefix out of range for this index";
                delete ix;
                return LM_ERR_OPTION;
            }
        }
        // local genomes
        int64_t nlocal = 0;
        for (int64_t g = 0; g < spec->genomes; g++)
            if ((int)(g % h.shard_count) == h.shard_rank) nlocal++;
        SynthDev sp;
        sp.seed = (uint64_t)spec->seed;
        sp.genomes = spec->genomes;
        sp.genome_len = spec->genome_len;
        sp.families = (int32_t)std::min<int64_t>(spec->families, spec->genomes);
        sp.max_div = spec->max_div;
        sp.shard_rank = h.shard_rank;
        sp.shard_count = h.shard_count;
        sp.nlocal = nlocal;
        sp.nblk = (spec->genome_len + 511) >> 9;
        sp.gbytes = ((((int64_t)spec->genome_len + 3) >> 2) + 16 + 7) & ~(int64_t)7;
        h.genome_batches = (int)((spec->genomes + 4999) / 5000);
        h.batch_first.assign(h.genome_batches + 1, 0);
        for (int b = 0; b <= h.genome_batches; b++) h.batch_first[b] = std::min<int64_t>((int64_t)b * 5000, spec->genomes);

        auto copy_up = [&](auto &dbuf, const auto &vec) {
            dbuf.ensure(std::max<size_t>(vec.size(), 1));
            HIPCHK(hipMemcpyAsync(dbuf.p, vec.data(), vec.size() * sizeof(vec[0]), hipMemcpyHostToDevice, ix->st));
        };
        copy_up(ix->d_masks, h.masks);
        copy_up(ix->d_pfx_first, pfx);
        copy_up(ix->d_batch_first, h.batch_first);
        // genomes
        ix->d_gbits.alloc_exact((size_t)(nlocal * sp.gbytes) + 64);
        HIPCHK(hipMemsetAsync(ix->d_gbits.p, 0, (size_t)(nlocal * sp.gbytes) + 64, ix->st));
        DBuf<int16_t> shifts;
        shifts.ensure((size_t)(nlocal * sp.nblk) + 1);
        hipLaunchKernelGGL(k_synth_shifts, dim3(gridn(nlocal, 64)), dim3(64), 0, ix->st, sp, shifts.p);
        hipLaunchKernelGGL(k_synth_genomes, dim3(gridn(nlocal * (((int64_t)spec->genome_len + 3) >> 2))), dim3(256), 0, ix->st,
                           sp, shifts.p, ix->d_gbits.p);
        bsync(ix);
        shifts.release();
        MaskTab mt{ix->d_masks.p, ix->d_pfx_first.p, K, p, M};
        // genome tables + host metadata
        std::vector<int64_t> goff(nlocal);
        std::vector<int32_t> glen(nlocal, spec->genome_len);
        std::vector<uint64_t> gbg(nlocal);
        h.genomes.resize(nlocal);
        for (int64_t l = 0; l < nlocal; l++) {
            int64_t g = h.shard_count > 1 ? l * h.shard_count + h.shard_rank : l;
            HostGenome &G = h.genomes[l];
            G.bg = ((uint64_t)(g / 5000) << 17) | (uint64_t)(g % 5000);
            G.global = g;
            char nm[64];
            snprintf(nm, sizeof nm, "SYN_%09lld.1", (long long)g);
            G.id = nm;
            G.genome_size = spec->genome_len;
            G.len = spec->genome_len;
            G.nseqs = 1;
            G.seq_sizes = {spec->genome_len};
            snprintf(nm, sizeof nm, "syn%09lld_c1", (long long)g);
            G.seq_ids = {std::string(nm)};
            G.bits_off = l * sp.gbytes;
            goff[l] = G.bits_off;
            gbg[l] = G.bg;
            ix->bg2local[G.bg] = (int)l;
        }
        copy_up(ix->d_g_off, goff);
        copy_up(ix->d_g_len, glen);
        copy_up(ix->d_g_bg, gbg);
        lm_fill_gap_lut(ix); // same table as lm_index_open (lib-chaining.go:662-667)
        bsync(ix);
        DevIndexView &v = ix->view;
        v.K = K;
        v.M = M;
        v.mask_prefix = p;
        v.masks = ix->d_masks.p;
        v.pfx_first = ix->d_pfx_first.p;
        v.g_bg = ix->d_g_bg.p;
        v.gbits = ix->d_gbits.p;
        v.g_off = ix->d_g_off.p;
        v.g_len = ix->d_g_len.p;
        v.batch_first = ix->d_batch_first.p;
        v.nbatches = h.genome_batches;
        v.ngenomes = nlocal;
        v.shard_rank = h.shard_rank;
        v.shard_count = h.shard_count;

        // ---- seeds: generated per chunk of genomes into a staging buffer and shown to the packer, twice (count, place):
        // the unpacked seeds of the whole set never exist (they would not fit beside the packed image at BASELINE configs 3-5)
        const bool dbg = getenv("LM_DEBUG") != nullptr;
        // capture in LDS when the per-mask minima fit a CU's LDS and the mask set has the lexicmap structure (every
        // p-base prefix once or twice); otherwise minima in a global table
        const int npfx = 1 << (2 * p), nwords = (npfx + 63) >> 6;
        bool lds_capture = true;
        std::vector<uint64_t> dmap(nwords, 0);
        std::vector<uint32_t> dcnt(nwords, 0);
        for (int f = 0; f < npfx; f++) {
            const int c = pfx[f + 1] - pfx[f];
            if (c < 1 || c > 2) lds_capture = false;
            if (c == 2) dmap[f >> 6] |= 1ull << (f & 63);
        }
        for (int w = 1; w < nwords; w++) dcnt[w] = dcnt[w - 1] + (uint32_t)__builtin_popcountll(dmap[w - 1]);
        const size_t lds_bytes = (size_t)M * 8 + (size_t)nwords * 12 + 64;
        if (lds_bytes > 160 * 1024) lds_capture = false;
        DBuf<uint64_t> dbl_map;
        DBuf<uint32_t> dbl_cnt;
        copy_up(dbl_map, dmap);
        copy_up(dbl_cnt, dcnt);
        if (lds_capture && lds_bytes > 64 * 1024)
            HIPCHK(hipFuncSetAttribute((const void *)k_capture_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        const int CH = (int)std::min<int64_t>(nlocal, lds_capture ? 2048 : std::max<int64_t>(1, (int64_t)(192ll << 20) / ((int64_t)M * 8)));
        double per_genome = 2.0 * (1.45 * M + (double)spec->genome_len / 42.0) + 1024;
        unsigned long long cap = (unsigned long long)(per_genome * (double)CH) + 65536;
        DBuf<uint16_t> s_mask;
        DBuf<uint64_t> s_kmer, s_val;
        s_mask.alloc_exact(cap);
        s_kmer.alloc_exact(cap);
        s_val.alloc_exact(cap);
        DBuf<unsigned long long> counters;
        counters.ensure(8);
        DBuf<unsigned long long> hashes;
        hashes.alloc_exact(lds_capture ? 1 : (size_t)CH * M);
        unsigned long long pos_cap = (unsigned long long)((1.45 * M + 64) * CH) + CH + 64;
        DBuf<uint64_t> pos_keys, pos_keys2;
        pos_keys.alloc_exact(pos_cap);
        pos_keys2.alloc_exact(pos_cap);
        const int64_t npos = (int64_t)spec->genome_len - K + 1;
        SeedPacker packer;
        packer.begin(ix, nlocal, spec->genome_len);
        double t_cap = 0, t_desert = 0, t_pack = 0;
        for (int pass = 0; pass < 2; pass++) {
            for (int64_t l0 = 0; l0 < nlocal; l0 += CH) {
                int nch = (int)std::min<int64_t>(CH, nlocal - l0);
                double ta = now_ms();
                HIPCHK(hipMemsetAsync(counters.p, 0, 2 * sizeof(unsigned long long), ix->st));
                if (lds_capture) {
                    hipLaunchKernelGGL(k_capture_lds, dim3(nch), dim3(1024), lds_bytes, ix->st, sp, mt, ix->d_gbits.p, l0,
                                       dbl_map.p, dbl_cnt.p, s_mask.p, s_kmer.p, s_val.p, counters.p, cap, pos_keys.p,
                                       counters.p + 1, pos_cap - CH - 1);
                } else {
                    hipLaunchKernelGGL(k_fill_u64, dim3(gridn((int64_t)nch * M)), dim3(256), 0, ix->st, hashes.p,
                                       (int64_t)nch * M, ~0ull);
                    hipLaunchKernelGGL(k_cap_argmin, dim3(gridn((int64_t)nch * npos)), dim3(256), 0, ix->st, sp, mt,
                                       ix->d_gbits.p, l0, nch, hashes.p);
                    hipLaunchKernelGGL(k_cap_emit, dim3(gridn((int64_t)nch * npos)), dim3(256), 0, ix->st, sp, mt, ix->d_gbits.p,
                                       l0, nch, hashes.p, s_mask.p, s_kmer.p, s_val.p, counters.p, cap, pos_keys.p,
                                       counters.p + 1, pos_cap - CH - 1);
                }
                unsigned long long hc[2];
                HIPCHK(hipMemcpyAsync(hc, counters.p, sizeof hc, hipMemcpyDeviceToHost, ix->st));
                bsync(ix);
                if (hc[0] >= cap || hc[1] >= pos_cap - CH - 1) throw HipError("synthetic builder: seed buffer too small");
                double tb = now_ms();
                unsigned long long npk = hc[1];
                hipLaunchKernelGGL(k_pseudo_pos, dim3((nch + 63) / 64), dim3(64), 0, ix->st, nch, (int32_t)(spec->genome_len - K),
                                   pos_keys.p, npk);
                npk += nch;
                prim_sort_keys(ix->st, ix->tmp, pos_keys.p, pos_keys2.p, (size_t)npk, 0, 64);
                hipLaunchKernelGGL(k_desert_fill, dim3(gridn(((int64_t)npk + 63) / 64, 4)), dim3(256), 0, ix->st, sp, mt, ix->d_gbits.p, l0,
                                   pos_keys2.p, (int64_t)npk, spec->max_desert, spec->seed_dist, s_mask.p, s_kmer.p, s_val.p,
                                   counters.p, cap);
                HIPCHK(hipMemcpyAsync(hc, counters.p, sizeof hc, hipMemcpyDeviceToHost, ix->st));
                bsync(ix);
                if (hc[0] >= cap) throw HipError("synthetic builder: seed buffer too small (desert)");
                unsigned long long upto = hc[0];
                hipLaunchKernelGGL(k_reverse_seeds, dim3(gridn((int64_t)upto)), dim3(256), 0, ix->st, mt, 0ull, upto, s_mask.p,
                                   s_kmer.p, s_val.p, counters.p, cap);
                HIPCHK(hipMemcpyAsync(hc, counters.p, sizeof hc, hipMemcpyDeviceToHost, ix->st));
                bsync(ix);
                if (hc[0] >= cap) throw HipError("synthetic builder: seed buffer too small (reversed)");
                double tc = now_ms();
                if (pass == 0)
                    packer.count(s_mask.p, s_kmer.p, s_val.p, (int64_t)hc[0]);
                else
                    packer.place(s_mask.p, s_kmer.p, s_val.p, (int64_t)hc[0]);
                bsync(ix);
                t_cap += tb - ta;
                t_desert += tc - tb;
                t_pack += now_ms() - tc;
            }
            if (pass == 0) packer.end_count();
            if (dbg)
                fprintf(stderr, "[lm] builder pass %d: capture %.0f ms, desert+reverse %.0f ms, packer %.0f ms (cumulative)\n", pass,
                        t_cap, t_desert, t_pack);
        }
        hashes.release();
        pos_keys.release();
        pos_keys2.release();
        s_mask.release();
        s_kmer.release();
        s_val.release();
        double tf = now_ms();
        packer.finish();
        if (dbg) fprintf(stderr, "[lm] builder: partition sort %.0f ms; %lld seeds (%lld outliers), %.2f B/seed\n", now_ms() - tf,
                         (long long)ix->n_seeds, (long long)ix->n_seeds_outlier, (double)ix->seed_bytes / std::max<double>(1.0, (double)ix->n_seeds));
        ix->tmp.release();
        ix->hbm_bytes = ix->seed_bytes + (int64_t)((uint64_t)(nlocal * sp.gbytes) + 64 + (uint64_t)M * 8 + pfx.size() * 4 + nlocal * 20);
        lm_set_scratch_budget(ix);
    } catch (const std::exception &e) {
        g_open_error = e.what();
        delete ix;
        return LM_ERR_HIP;
    }
    *out = ix;
    return LM_OK;
}

lm_status lm_index_fetch(lm_index *ix, int64_t local_genome, int64_t start, int64_t len, uint8_t *out) {
    if (!ix || local_genome < 0 || local_genome >= (int64_t)ix->host.genomes.size()) return LM_ERR_ARG;
    const HostGenome &G = ix->host.genomes[local_genome];
    if (start < 0 || len < 0 || start + len > G.len) return LM_ERR_ARG;
    try {
        std::lock_guard<std::mutex> lock(ix->mu);
        HIPCHK(hipSetDevice(ix->device));
        DBuf<uint8_t> d;
        d.ensure((size_t)len + 1);
        hipLaunchKernelGGL(k_fetch_bases, dim3(gridn(len)), dim3(256), 0, ix->st, ix->d_gbits.p + G.bits_off, start, len, d.p);
        HIPCHK(hipMemcpyAsync(out, d.p, (size_t)len, hipMemcpyDeviceToHost, ix->st));
        bsync(ix);
    } catch (const std::exception &e) {
        ix->err = e.what();
        return LM_ERR_HIP;
    }
    return LM_OK;
}

} // extern "C"
