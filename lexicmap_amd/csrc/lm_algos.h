// lm_algos.h — per-work-item device algorithms of the `lexicmap search` hot path (gfx950 / CDNA4).
//
// Everything here is LM_HD (__host__ __device__) and works on caller-provided scratch: the HIP kernels in
// lm_kernels.hip call these from one lane (or one wave) per work item.  They are also compiled for the host by
// tests/host_algos.cpp so that the `-m "not gpu"` suite can check the device logic against the oracle; the product
// library never executes them on the host.
//
// Reference citations are relative to /root/reference/lexicmap/cmd.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define LM_HD __host__ __device__ __forceinline__
#define LM_HDN __host__ __device__ inline
#else
#define LM_HD inline
#define LM_HDN inline
#endif

#define LM_NULL_OFF (-1073741824) /* INT32_MIN/2 */

// ---------------------------------------------------------------------------------------------------------------
// k-mers (util/kmers.go, genome/genome.go:1427-1444)
LM_HD int lm_clz64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return x ? __clzll((long long)x) : 64;
#else
    return x ? __builtin_clzll(x) : 64;
#endif
}

LM_HD uint8_t lm_base2bit(uint8_t c) {
    // A=0 C=1 G=2 T/U=3; B,S,Y->1; K->2; everything else 0 (case-insensitive)
    switch (c | 0x20) {
    case 'c': case 'b': case 's': case 'y': return 1;
    case 'g': case 'k': return 2;
    case 't': case 'u': return 3;
    default: return 0;
    }
}

LM_HD uint64_t lm_kmer_mask(int k) { return k >= 32 ? ~0ull : ((1ull << (k << 1)) - 1); }

// reverse the order of the 32 2-bit groups of a word: byte swap (v_perm on gfx950), then the two sub-byte stages
LM_HD uint64_t lm_reverse_groups(uint64_t x) {
    x = __builtin_bswap64(x);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    return x;
}
LM_HD uint64_t lm_revcomp(uint64_t x, int k) { // complement, reverse the bases
    return lm_reverse_groups(~x) >> (64 - (k << 1));
}

// kmers.MustReverse: base-wise reversal without complement (lib-index-search.go:1327)
LM_HD uint64_t lm_reverse(uint64_t x, int k) { return lm_reverse_groups(x) >> (64 - (k << 1)); }

LM_HD uint64_t lm_ns(uint64_t b, int k) {
    uint64_t c = b;
    for (int i = 1; i < k; i++) c = (c << 2) + b;
    return c;
}

// util/kmers.go:162-328 — DUST over the k-1 overlapping 3-mers (the last window reaches two zero bits above the k-mer)
LM_HD bool lm_dust(uint64_t code, int k) {
    // 64 bins of <=31 counts: 5 bits each would do, use 8-bit lanes in eight u64 words
    uint64_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i <= k - 2; i++) {
        uint32_t b = (uint32_t)(code >> (i << 1)) & 63u;
        cnt[b >> 3] += 1ull << ((b & 7) << 3);
    }
    uint32_t score = 0;
    for (int w = 0; w < 8; w++) {
        uint64_t x = cnt[w];
        for (int j = 0; j < 8; j++) {
            uint32_t c = (uint32_t)(x & 0xff);
            x >>= 8;
            score += (c * (c - 1)) >> 1; // c=0 -> 0
        }
    }
    return (score & 0xffffu) > 50u; // uint16 arithmetic in the reference; the sum is < 2^16 for k<=32
}

// lib-index-search.go:1223-1238
LM_HD bool lm_low_complexity(uint64_t kmer, int k) {
    return kmer == lm_ns(1, k) || kmer == lm_ns(2, k) || kmer == lm_kmer_mask(k) || lm_dust(kmer, k);
}

LM_HD int lm_lcp(uint64_t a, uint64_t b, int k) { return (lm_clz64(a ^ b) >> 1) + k - 32; }

// ---------------------------------------------------------------------------------------------------------------
// Packed seed image: the seeds of one (mask, direction) whose k-mers start with the mask's p-base prefix
// are bucketed by their next a bases (the reference's anchor partitions, kv-data.go:90-125) and stored as two bit
// streams of fixed-width elements, little-endian bit order inside 64-bit words:
//   key  = the k-mer's last K-p-a bases                      (key_bits = 2 (K-p-a); 36 for K=31, p=7, a=6)
//   val  = local genome number | position | strand           (gid_bits + pos_bits + 1; 40 for 100k x 3-Mb genomes)
// versus the 2 x 64 bits of the reference's RAM form (kv-reader.go:762-1021): k-mer u64 + value u64
// (batch:17|genome:17|pos:28|rc:1|reversed:1, lib-index-build.go:412-455).  The reversed flag is the direction of the
// list, the p-base prefix is the mask's, the a-base partition is the table index.
LM_HD uint64_t lm_bits_get(const uint64_t *a, int64_t i, int w) {
    const int64_t bit = i * (int64_t)w;
    const int64_t word = bit >> 6;
    const int sh = (int)(bit & 63);
    uint64_t v = a[word] >> sh;
    if (sh + w > 64) v |= a[word + 1] << (64 - sh);
    return w >= 64 ? v : (v & ((1ull << w) - 1));
}
LM_HD uint64_t lm_pack_seed_val(uint64_t local_genome, uint64_t v64, int pos_bits) {
    const uint64_t pos = (v64 >> 2) & 0xfffffffull, rc = (v64 >> 1) & 1ull;
    return (local_genome << (pos_bits + 1)) | (pos << 1) | rc;
}
// back to the reference's value layout given the genome's batch:17|genome:17 key and the list's direction
LM_HD uint64_t lm_unpack_seed_val(uint64_t pv, uint64_t bg, int pos_bits, int dir) {
    const uint64_t pos = (pv >> 1) & ((1ull << pos_bits) - 1);
    return (bg << 30) | (pos << 2) | ((pv & 1ull) << 1) | (uint64_t)dir;
}
LM_HD uint64_t lm_packed_val_genome(uint64_t pv, int pos_bits) { return pv >> (pos_bits + 1); }
// word `w` of a bit stream rebuilt from the elements [first, first+n) that overlap it (get(i) = element first+i);
// *mask = the bits of the word that belong to those elements (the rest belongs to the neighbouring partitions)
template <typename Get>
LM_HD uint64_t lm_bits_build_word(int64_t w, int64_t first, int64_t n, int width, Get get, uint64_t *mask) {
    const int64_t wlo = w << 6, whi = wlo + 64; // bit range of the word
    int64_t e0 = wlo / width, e1 = (whi - 1) / width;
    if (e0 < first) e0 = first;
    if (e1 > first + n - 1) e1 = first + n - 1;
    uint64_t v = 0, m = 0;
    const uint64_t em = width >= 64 ? ~0ull : ((1ull << width) - 1);
    for (int64_t e = e0; e <= e1; e++) {
        const int64_t b = e * width - wlo; // negative: the element starts in the previous word
        const uint64_t x = get(e - first);
        if (b >= 0) {
            v |= x << b;
            m |= em << b;
        } else {
            v |= x >> (-b);
            m |= em >> (-b);
        }
    }
    *mask = m;
    return v;
}
// seeds of the sorted partition [b, e) whose key lies in [lrem, rrem]: returns the count, *first = the first one
// (kv-searcher2.go:105-323: lower bound of the range's left end, then scan)
LM_HD int32_t lm_partition_range(const uint64_t *keys, int key_bits, int64_t b, int64_t e, uint64_t lrem, uint64_t rrem,
                                 int64_t *first) {
    int64_t lo = b, hi = e;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (lm_bits_get(keys, mid, key_bits) < lrem)
            lo = mid + 1;
        else
            hi = mid;
    }
    *first = lo;
    while (lo < e && lm_bits_get(keys, lo, key_bits) <= rrem) lo++;
    return (int32_t)(lo - *first);
}

// ---------------------------------------------------------------------------------------------------------------
// LexicHash capture of one mask over a query's sorted k-mer array (lexichash.Mask semantics: argmin of mask^kmer).
// a[0..n) sorted ascending (duplicates allowed). Returns the winning k-mer; [*lo,*hi) = its occurrences.
LM_HD uint64_t lm_xor_argmin(const uint64_t *a, int n, uint64_t m, int *lo_out, int *hi_out) {
    int lo = 0, hi = n;
    while (true) {
        uint64_t x = a[lo], y = a[hi - 1];
        if (x == y) break;
        int b = 63 - lm_clz64(x ^ y); // highest bit that differs inside the range
        // first index whose bit b is set (elements share all higher bits, so bit b is monotone)
        int l = lo, h = hi;
        while (l < h) {
            int mid = (l + h) >> 1;
            if ((a[mid] >> b) & 1)
                h = mid;
            else
                l = mid + 1;
        }
        if ((m >> b) & 1)
            lo = l;
        else
            hi = l;
    }
    *lo_out = lo;
    *hi_out = hi;
    return a[lo];
}

// ---------------------------------------------------------------------------------------------------------------
// Seed-value decode + anchor coordinates (lib-index-search.go:1491-1524)
LM_HD void lm_anchor_coords(uint64_t refpos, int posq, bool rcq, int kprefix, int K, int *beginq, int *begint,
                            bool *rct_out) {
    int post = (int)((refpos << 34) >> 36);
    bool rvt = (refpos & 1) != 0;
    bool rct = ((refpos >> 1) & 1) != 0;
    if (!rvt) {
        *beginq = rcq ? posq + K - kprefix : posq;
        *begint = rct ? post + K - kprefix : post;
    } else {
        *beginq = rcq ? posq : posq + K - kprefix;
        *begint = rct ? post : post + K - kprefix;
    }
    *rct_out = rct;
}

// Anchor sort key: (QBegin asc, QEnd desc, TBegin asc, QRC asc, TRC asc) — the total order used in place of the
// reference's unstable slices.SortFunc (lib-index-search.go:868-876).  27|6|29|1|1 bits.
LM_HD uint64_t lm_pack_anchor(int qbegin, int len, int tbegin, bool qrc, bool trc) {
    return ((uint64_t)(uint32_t)qbegin << 37) | ((uint64_t)(uint32_t)(32 - len) << 31) |
           ((uint64_t)(uint32_t)tbegin << 2) | ((uint64_t)(qrc ? 1 : 0) << 1) | (uint64_t)(trc ? 1 : 0);
}

struct LmSub { // SubstrPair, lib-index-search.go:805-817
    int32_t qbegin, tbegin;
    uint8_t len, trc, qrc, pad;
};

LM_HD LmSub lm_unpack_anchor(uint64_t v) {
    LmSub s;
    s.qbegin = (int32_t)(v >> 37);
    s.len = (uint8_t)(32 - (int)((v >> 31) & 63));
    s.tbegin = (int32_t)((v >> 2) & 0x1fffffffu);
    s.qrc = (uint8_t)((v >> 1) & 1);
    s.trc = (uint8_t)(v & 1);
    s.pad = 0;
    return s;
}

// ---------------------------------------------------------------------------------------------------------------
// ClearSubstrPairs on an already sorted list (lib-index-search.go:878-990). Compacts in place, returns new n.
// marks: scratch [n] bytes.
LM_HDN int lm_clear_sorted(LmSub *subs, int n, int k, uint8_t *marks) {
    if (n <= 1) return n;
    for (int i = 0; i < n; i++) marks[i] = 0;
    for (int i = 0; i + 1 < n; i++) {
        const LmSub v = subs[i + 1];
        int32_t vqend = v.qbegin + v.len;
        int32_t upbound = vqend - k;
        if (upbound < 0) upbound = 0;
        int32_t vtbegin = v.tbegin, vtend = v.tbegin + v.len;
        int lo = 0, hi = i + 1;
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if (subs[mid].qbegin < upbound)
                lo = mid + 1;
            else
                hi = mid;
        }
        for (int j = lo; j <= i; j++) {
            const LmSub p = subs[j];
            if (vqend <= p.qbegin + p.len && vtbegin >= p.tbegin && vtend <= p.tbegin + p.len) {
                marks[i + 1] = 1;
                break;
            }
        }
    }
    int j = 0;
    for (int i = 0; i < n; i++)
        if (!marks[i]) subs[j++] = subs[i];
    return j;
}

// ---------------------------------------------------------------------------------------------------------------
// in-place heapsort of u64 (used for the range index and the score list of Chainer; n is small)
LM_HDN void lm_heapsort_u64(uint64_t *a, int n) {
    for (int start = n / 2 - 1; start >= 0; start--) {
        int root = start;
        uint64_t v = a[root];
        while (true) {
            int child = 2 * root + 1;
            if (child >= n) break;
            if (child + 1 < n && a[child] < a[child + 1]) child++;
            if (v >= a[child]) break;
            a[root] = a[child];
            root = child;
        }
        a[root] = v;
    }
    for (int end = n - 1; end > 0; end--) {
        uint64_t v = a[end];
        a[end] = a[0];
        int root = 0;
        while (true) {
            int child = 2 * root + 1;
            if (child >= end) break;
            if (child + 1 < end && a[child] < a[child + 1]) child++;
            if (v >= a[child]) break;
            a[root] = a[child];
            root = child;
        }
        a[root] = v;
    }
}

LM_HD uint32_t lm_f32bits(float f) {
    union { float f; uint32_t u; } x;
    x.f = f;
    return x.u;
}
LM_HD float lm_f32frombits(uint32_t u) {
    union { float f; uint32_t u; } x;
    x.u = u;
    return x.f;
}

// seedWeight (lib-chaining.go:635). Compiled with -ffp-contract=off: two separately rounded float32 multiplies.
LM_HD float lm_seed_weight(float l) { return 0.1f * l * l; }

struct LmChainOpt {       // ChainingOptions, lib-index-search.go:739-746
    float max_gap;        // --seed-max-gap
    float min_score;      // seedWeight(MinSinglePrefix)
    float max_distance;   // --seed-max-dist
    int top_chains;       // -N
    int gap_lut_n;        // entries in gap_lut
    const float *gap_lut; // gapScore(g) for integer g in [0, gap_lut_n): host-built with Go's Log2 (appendix B.5)
};

// Chainer.Chain, lib-chaining.go:122-633.
// scratch: msi[n], s2i[n] (u64); dirs[n] (i8); visited[n] (u8).
// out: chain_off[0..nchains] (capacity n+2), chain_idx (capacity 2n+2). returns best score.
LM_HDN float lm_run_chain1(const LmSub *subs, int n, const LmChainOpt &opt, uint64_t *msi, uint64_t *s2i, int8_t *dirs,
                       uint8_t *visited, int32_t *chain_off, int32_t *chain_idx, int *nchains_out) {
    int nchains = 0, nidx = 0;
    chain_off[0] = 0;
    if (n == 1) {
        float w = lm_seed_weight((float)subs[0].len);
        if (w >= opt.min_score) {
            chain_idx[nidx++] = 0;
            chain_off[++nchains] = nidx;
        }
        *nchains_out = nchains;
        return w;
    }
    float s = lm_seed_weight((float)subs[0].len);
    msi[0] = (uint64_t)lm_f32bits(s) << 32;
    dirs[0] = 0;
    s2i[0] = (uint64_t)lm_f32bits(s) << 32;
    const int32_t max_dist_i = (int32_t)opt.max_distance;
    for (int i = 1; i < n; i++) {
        const LmSub a = subs[i];
        const int32_t aq = a.qbegin, alen = a.len;
        float m = lm_seed_weight((float)alen);
        int mj = i;
        int8_t mdir = 0;
        // Candidates: every j < i with |TBegin_j - TBegin_i| <= maxDistance (range index + sort of indices in the
        // reference, :380-399), scanned from high j to low j.  Since subs is sorted by QBegin the `break` at
        // :416-419 is equivalent to the filter QBegin_i - QBegin_j <= maxDistance, and "s > m" while scanning
        // downwards keeps the LARGEST j among equal best scores.  Scanning j = i-1 .. 0 directly with the TBegin
        // window test visits the same candidates in the same order.
        int64_t tlo = (int64_t)a.tbegin - max_dist_i;
        if (a.tbegin < max_dist_i) tlo = 0;
        int64_t thi = (int64_t)a.tbegin + max_dist_i;
        for (int j = i - 1; j >= 0; j--) {
            const LmSub b = subs[j];
            if (aq - b.qbegin > max_dist_i) break;
            if ((int64_t)b.tbegin < tlo || (int64_t)b.tbegin > thi) continue;
            if (a.qbegin == b.qbegin || a.tbegin == b.tbegin) continue;
            // gap(), lib-chaining.go:655-660 (integers; exact in float32)
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
            float g = (float)gi;
            if (g > opt.max_gap) continue;
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
            int8_t dir = a.tbegin >= b.tbegin ? 1 : -1;
            float gs = gi < opt.gap_lut_n ? opt.gap_lut[gi] : 0.0f;
            if (dirs[j] == 0 || dirs[j] == dir) {
                float t = lm_f32frombits((uint32_t)(msi[j] >> 32)) + w;
                s = t - gs;
            } else {
                float t = lm_seed_weight((float)b.len) + w;
                s = t - gs;
            }
            if (s >= opt.min_score && s > m) {
                m = s;
                mj = j;
                mdir = dir;
            }
        }
        msi[i] = ((uint64_t)lm_f32bits(m) << 32) | (uint32_t)mj;
        dirs[i] = mdir;
        s2i[i] = ((uint64_t)lm_f32bits(m) << 32) | (uint32_t)i;
    }
    // backtrack, :490-632
    for (int i = 0; i < n; i++) visited[i] = 0;
    lm_heapsort_u64(s2i, n);
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
            M = lm_f32frombits((uint32_t)(s2i[imax] >> 32));
            Mi = (uint32_t)s2i[imax];
            if (!visited[Mi]) {
                imax--;
                break;
            }
            imax--;
        }
        if (M < opt.min_score) break;
        int pstart = nidx; // path is written descending, reversed on success
        int i = (int)Mi;
        if (first) {
            max_score = M;
            first = false;
        }
        while (true) {
            int j = (int)(msi[i] & 4294967295ull);
            bool change = (i != j && dirs[j] != 0 && dirs[i] != dirs[j]);
            if (visited[j] && !change) {
                nidx = pstart; // path abandoned
                visited[i] = 1;
                break;
            }
            chain_idx[nidx++] = i;
            visited[i] = 1;
            if (i == j || change) {
                if (change) chain_idx[nidx++] = j;
                for (int x = pstart, y = nidx - 1; x < y; x++, y--) {
                    int32_t t = chain_idx[x];
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
    *nchains_out = nchains;
    return max_score;
}

// ---------------------------------------------------------------------------------------------------------------
// TrimSubStrPairs (lib-seq_compare.go:553-634). Returns new n; *start_out = first kept index.
LM_HD float lm_distance_f32(const LmSub &a, const LmSub &b) {
    int32_t x = a.qbegin - b.qbegin, y = a.tbegin - b.tbegin;
    if (x < 0) x = -x;
    if (y < 0) y = -y;
    return (float)(x > y ? x : y);
}
LM_HD int32_t lm_gap2i(const LmSub &a, const LmSub &b) {
    int32_t x = a.qbegin - b.qbegin, y = a.tbegin - b.tbegin;
    if (x < 0) x = -x;
    if (y < 0) y = -y;
    int32_t g = x - y;
    return g < 0 ? -g : g;
}
LM_HD int32_t lm_overlap(const LmSub &a, const LmSub &b) {
    int32_t qo = 0, to = 0;
    if (b.qbegin >= a.qbegin && b.qbegin <= a.qbegin + (int32_t)a.len) qo = a.qbegin + (int32_t)a.len - b.qbegin + 1;
    if (b.tbegin >= a.tbegin && b.tbegin <= a.tbegin + (int32_t)a.len) to = a.tbegin + (int32_t)a.len - b.tbegin + 1;
    return qo > to ? qo : to;
}
LM_HDN int lm_trim(const LmSub *subs, int n, float min_dist, int *start_out) {
    *start_out = 0;
    if (n < 2) return n;
    int last = n - 1;
    LmSub _p = subs[0];
    int start = 0;
    for (int i = 0; i < n - 1; i++) { // i indexes (*subs)[1:], the reference stores that index in `start`
        const LmSub p = subs[i + 1];
        if (lm_distance_f32(p, _p) < min_dist &&
            ((p.qbegin == _p.qbegin || p.tbegin == _p.tbegin) ||
             (lm_gap2i(_p, p) > 11 && (double)lm_overlap(_p, p) / (double)_p.len > 0.8))) {
            start = i;
            _p = p;
            continue;
        }
        break;
    }
    _p = subs[last];
    int end = last;
    for (int i = n - 2; i >= 0; i--) {
        const LmSub p = subs[i];
        if (lm_distance_f32(p, _p) < min_dist &&
            ((p.qbegin == _p.qbegin || p.tbegin == _p.tbegin) ||
             (lm_gap2i(p, _p) > 11 && (double)lm_overlap(p, _p) / (double)_p.len > 0.8))) {
            end = i;
            _p = p;
            continue;
        }
        break;
    }
    if (start >= end) return 0;
    *start_out = start;
    return end - start + 1;
}

// ---------------------------------------------------------------------------------------------------------------
// Chainer2 (lib-chaining2.go:152-658)
struct LmChain2Opt { // search.go:364-378
    int max_gap, min_score, min_align_len, band_count, band_base;
    double heuristic_pident;
};
struct LmChain2 { // the fields of Chain2Result produced by chaining
    int32_t qbegin, qend, tbegin, tend;
    int32_t nanchors, matched_bases, aligned_bases_q, aligned_bases_t;
    double pident;
};

// scratch: msi[n] (u64), stack[2*(n+1)] (i32). out: chains (capacity n). returns #chains in emission order
// (chain, then right region, then left region — the recursion order of chainARegion).
LM_HDN int lm_run_chain2(const LmSub *subs, int n, const LmChain2Opt &opt, uint64_t *msi, int32_t *stack, LmChain2 *out) {
    if (n <= 0) return 0;
    if (n == 1) { // :155-180
        int slen = subs[0].len;
        if (slen >= opt.min_score && slen >= opt.min_align_len) {
            LmChain2 p;
            p.qbegin = subs[0].qbegin;
            p.qend = subs[0].qbegin + slen - 1;
            p.tbegin = subs[0].tbegin;
            p.tend = subs[0].tbegin + slen - 1;
            p.matched_bases = slen;
            p.pident = 100;
            p.aligned_bases_q = slen;
            p.aligned_bases_t = 0;
            p.nanchors = 1;
            out[0] = p;
            return 1;
        }
        return 0;
    }
    const int32_t band_base = opt.band_base;
    const int band_count = opt.band_count;
    msi[0] = (uint64_t)subs[0].len << 32;
    // scores are integer-valued doubles in the reference (float64(len) - g ...): int64 arithmetic is exact
    int64_t M = 0;
    int Mi = 0;
    for (int i = 1; i < n; i++) {
        const LmSub a = subs[i];
        int64_t m = a.len;
        int mj = i;
        const int32_t aq = a.qbegin, at = a.tbegin;
        int bcount = 0;
        for (int j = i - 1; j >= 0; j--) {
            const LmSub b = subs[j];
            const int32_t bq = b.qbegin, bt = b.tbegin;
            if (bq == aq || bt > at) continue;
            bcount++;
            int32_t bbase = aq - bq - (int32_t)b.len;
            if (!(bbase <= band_base || bcount <= band_count)) break;
            int32_t qd = aq - bq, td = at - bt;
            if (qd < 0) qd = -qd;
            if (td < 0) td = -td;
            int32_t g = qd > td ? qd - td : td - qd;
            if (g > opt.max_gap) continue;
            int64_t s = (int64_t)(msi[j] >> 32) + (int64_t)b.len - (int64_t)g;
            if (s >= m) {
                m = s;
                mj = j;
            }
        }
        // uint64(m)<<32 | uint64(mj): m can be negative only if ... it cannot: m >= len > 0
        msi[i] = ((uint64_t)m << 32) | (uint64_t)(uint32_t)mj;
        if (m > M) {
            M = m;
            Mi = i;
        }
    }
    if (M < (int64_t)opt.min_score) return 0;
    int nout = 0;
    int sp = 0;
    // frame = (lo, hi, Mi0) with Mi0 = -1 when the maximum must be searched
    stack[sp++] = 0;
    stack[sp++] = n;
    int pending_Mi0 = Mi;
    while (sp > 0) {
        int hi = stack[--sp];
        int lo = stack[--sp];
        int mi;
        if (pending_Mi0 >= 0) {
            mi = pending_Mi0;
            pending_Mi0 = -1;
        } else {
            int64_t best = 0;
            mi = lo;
            for (int i = lo; i < hi; i++) {
                int64_t m = (int64_t)(msi[i] >> 32);
                if (m > best) {
                    best = m;
                    mi = i;
                }
            }
            if (best < (int64_t)opt.min_score) continue;
        }
        int n_matched = 0, n_abq = 0, n_abt = 0;
        int i = mi, j = 0;
        int32_t qb = 0, qe = 0, tb = 0, te = 0;
        int begin_of_next = 0;
        bool first_anchor = true;
        int n_anchors = 0;
        bool jneg = false;
        while (true) {
            j = (int)(msi[i] & 4294967295ull);
            if (j < lo) {
                jneg = true;
                break;
            }
            const LmSub sub = subs[i];
            n_anchors++;
            if (first_anchor) {
                first_anchor = false;
                qe = sub.qbegin + (int32_t)sub.len - 1;
                te = sub.tbegin + (int32_t)sub.len - 1;
                qb = sub.qbegin;
                tb = sub.tbegin;
                n_matched += sub.len;
            } else {
                qb = sub.qbegin;
                tb = sub.tbegin;
                if ((int)sub.qbegin + (int)sub.len - 1 >= begin_of_next)
                    n_matched += begin_of_next - (int)sub.qbegin;
                else
                    n_matched += sub.len;
            }
            begin_of_next = sub.qbegin;
            if (i == j) {
                n_abq += (int)qe - (int)qb + 1;
                if (n_abq < opt.min_align_len) break;
                n_abt += (int)te - (int)tb + 1;
                double pident = (double)n_matched / (double)(n_abq > n_abt ? n_abq : n_abt) * 100;
                if (pident < opt.heuristic_pident) break;
                if (pident > 100) pident = 100;
                LmChain2 p;
                p.nanchors = n_anchors;
                p.aligned_bases_q = n_abq;
                p.aligned_bases_t = n_abt;
                p.matched_bases = n_matched;
                p.pident = pident;
                p.qbegin = qb;
                p.qend = qe;
                p.tbegin = tb;
                p.tend = te;
                out[nout++] = p;
                break;
            }
            i = j;
        }
        if (jneg && n_anchors > 0) { // :534-569
            n_abq += (int)qe - (int)qb + 1;
            n_abt += (int)te - (int)tb + 1;
            if (n_abq >= opt.min_align_len) {
                double pident = (double)n_matched / (double)(n_abq > n_abt ? n_abq : n_abt) * 100;
                if (pident >= opt.heuristic_pident) {
                    if (pident > 100) pident = 100;
                    LmChain2 p;
                    p.nanchors = n_anchors;
                    p.aligned_bases_q = n_abq;
                    p.aligned_bases_t = n_abt;
                    p.matched_bases = n_matched;
                    p.pident = pident;
                    p.qbegin = qb;
                    p.qend = qe;
                    p.tbegin = tb;
                    p.tend = te;
                    out[nout++] = p;
                }
            }
        }
        // the reference recurses right ([mi+1,hi)) first, then left ([lo,i)); push left first so right pops first
        if (i > lo) {
            stack[sp++] = lo;
            stack[sp++] = i;
        }
        if (mi != hi - 1) {
            stack[sp++] = mi + 1;
            stack[sp++] = hi;
        }
    }
    return nout;
}

// ---------------------------------------------------------------------------------------------------------------
// Chainer3 (lib-chaining3.go:111-299) with DefaultChaining3Options. scratch msi[n] (i64). returns found.
LM_HDN bool lm_run_chain3(const LmSub *subs, int n, int64_t *msi, int *qend_out, int *tend_out) {
    const int32_t band_base = 10;
    const int band_count = 20;
    const int64_t max_gap = 5, max_distance = 10, min_score = 1;
    const int min_align_len = 2;
    if (n <= 0) return false;
    int64_t M = 0;
    int Mi = 0;
    for (int i = 0; i < n; i++) {
        const LmSub a = subs[i];
        // m = len - distance2(sub0,a) - gap2(sub0,a), sub0 = (0,0)
        int64_t aq = a.qbegin < 0 ? -(int64_t)a.qbegin : a.qbegin, at = a.tbegin < 0 ? -(int64_t)a.tbegin : a.tbegin;
        int64_t m = (int64_t)a.len - (aq > at ? aq : at) - (aq > at ? aq - at : at - aq);
        int mj = i;
        if (i > 0) {
            int bcount = 0;
            for (int j = i - 1; j >= 0; j--) {
                const LmSub b = subs[j];
                if (b.qbegin == a.qbegin || b.tbegin > a.tbegin) continue;
                bcount++;
                int32_t bbase = a.qbegin - b.qbegin - (int32_t)b.len;
                if (!(bbase <= band_base || bcount <= band_count)) break;
                int64_t dq = a.qbegin - b.qbegin, dt = a.tbegin - b.tbegin;
                if (dq < 0) dq = -dq;
                if (dt < 0) dt = -dt;
                int64_t d = dq > dt ? dq : dt;
                if (d > max_distance) continue;
                int64_t g = dq > dt ? dq - dt : dt - dq;
                if (g > max_gap) continue;
                int64_t s = (msi[j] >> 32) + (int64_t)b.len - d - g;
                if (s >= m) {
                    m = s;
                    mj = j;
                }
            }
        }
        msi[i] = (int64_t)(((uint64_t)m << 32) | (uint64_t)(uint32_t)mj);
        if (i > 0 && m > M) {
            M = m;
            Mi = i;
        }
    }
    if (M < min_score) return false;
    int n_matched = 0, n_abq = 0, n_abt = 0;
    int i = Mi;
    int32_t qb = 0, qe = 0, tb = 0, te = 0;
    int begin_of_next = 0;
    bool first_anchor = true;
    while (true) {
        int j = (int)(msi[i] & 4294967295ll);
        const LmSub sub = subs[i];
        if (first_anchor) {
            first_anchor = false;
            qe = sub.qbegin + (int32_t)sub.len - 1;
            te = sub.tbegin + (int32_t)sub.len - 1;
            qb = sub.qbegin;
            tb = sub.tbegin;
            n_matched += sub.len;
        } else {
            qb = sub.qbegin;
            tb = sub.tbegin;
            if ((int)sub.qbegin + (int)sub.len - 1 >= begin_of_next)
                n_matched += begin_of_next - (int)sub.qbegin;
            else
                n_matched += sub.len;
        }
        begin_of_next = sub.qbegin;
        if (i == j) {
            n_abq += (int)qe - (int)qb + 1;
            if (n_abq < min_align_len) return false;
            n_abt += (int)te - (int)tb + 1;
            double pident = (double)n_matched / (double)(n_abq > n_abt ? n_abq : n_abt) * 100;
            if (pident < 15) return false;
            *qend_out = qe;
            *tend_out = te;
            return true;
        }
        i = j;
    }
}

// _extendRight (lib-index-search-util.go:98-201): 2-mer exact matches between the two flanks, sorted, Chainer3.
// rev=true reads both flanks backwards (the reference reverses copies for the 5' side).
// scratch: subs[cap], msi[cap]; flanks are at most ~130 bases => cap >= n1*n2/... callers size by (n1-1)*(n2-1).
LM_HD uint8_t lm_flank_base(const uint8_t *s, int n, int i, bool rev) { return rev ? s[n - 1 - i] : s[i]; }
LM_HDN void lm_extend_right(const uint8_t *s1, int n1, const uint8_t *s2, int n2, bool rev, LmSub *subs, int64_t *msi,
                            int cap, int *o1, int *o2) {
    *o1 = 0;
    *o2 = 0;
    if (n1 < 2 || n2 < 2) return;
    // Anchors in the order (QBegin asc, QEnd desc [all 2], TBegin asc): generate directly in that order.
    int n = 0;
    for (int p = 0; p + 1 < n1; p++) {
        uint32_t km1 = (lm_base2bit(lm_flank_base(s1, n1, p, rev)) << 2) | lm_base2bit(lm_flank_base(s1, n1, p + 1, rev));
        for (int t = 0; t + 1 < n2; t++) {
            uint32_t km2 =
                (lm_base2bit(lm_flank_base(s2, n2, t, rev)) << 2) | lm_base2bit(lm_flank_base(s2, n2, t + 1, rev));
            if (km1 == km2) {
                if (n >= cap) return; // cannot happen when cap >= (n1-1)*(n2-1)
                LmSub x;
                x.qbegin = p;
                x.tbegin = t;
                x.len = 2;
                x.qrc = x.trc = x.pad = 0;
                subs[n++] = x;
            }
        }
    }
    if (n == 0) return;
    int qe, te;
    if (lm_run_chain3(subs, n, msi, &qe, &te)) {
        *o1 = qe + 1;
        *o2 = te + 1;
    }
}

// extendMatch (lib-index-search-util.go:34-96). seq1=query (len1), seq2=target window (len2).
LM_HDN void lm_extend_match(const uint8_t *seq1, int len1, const uint8_t *seq2, int len2, int start1, int end1,
                            int start2, int end2, int ext_len, int tbegin, int max_ext_len, bool rc, LmSub *subs,
                            int64_t *msi, int cap, int *o_start1, int *o_end1, int *o_start2, int *o_end2, int *s1o,
                            int *e1o, int *s2o, int *e2o) {
    const int m = 2;
    int _start1 = start1, _end1 = end1, _start2 = start2, _end2 = end2;
    int _s1 = 0, _e1 = 0, _s2 = 0, _e2 = 0, _ext;
    if (end1 + m < len1 && end2 + m < len2) {
        _ext = rc ? (ext_len < tbegin ? ext_len : tbegin) : (ext_len < max_ext_len ? ext_len : max_ext_len);
        if (_ext > 2) {
            int e1 = end1 + _ext < len1 ? end1 + _ext : len1;
            int e2 = end2 + _ext < len2 ? end2 + _ext : len2;
            lm_extend_right(seq1 + end1, e1 - end1, seq2 + end2, e2 - end2, false, subs, msi, cap, &_e1, &_e2);
            if (_e1 > 0 || _e2 > 0) {
                end1 += _e1;
                end2 += _e2;
            }
        }
    }
    if (start1 > m && start2 > m) {
        _ext = rc ? (ext_len < max_ext_len ? ext_len : max_ext_len) : (ext_len < tbegin ? ext_len : tbegin);
        if (_ext > 2) {
            int s1 = start1 - _ext > 0 ? start1 - _ext : 0;
            int s2 = start2 - _ext > 0 ? start2 - _ext : 0;
            lm_extend_right(seq1 + s1, start1 - s1, seq2 + s2, start2 - s2, true, subs, msi, cap, &_s1, &_s2);
            if (_s1 > 0 || _s2 > 0) {
                start1 -= _s1;
                start2 -= _s2;
            }
        }
    }
    if (start1 < 0 || start2 < 0) {
        start1 = _start1;
        start2 = _start2;
    }
    if (end1 > len1 || end2 > len2) {
        end1 = _end1;
        end2 = _end2;
    }
    *o_start1 = start1;
    *o_end1 = end1;
    *o_start2 = start2;
    *o_end2 = end2;
    *s1o = _s1;
    *e1o = _e1;
    *s2o = _s2;
    *e2o = _e2;
}

// ---------------------------------------------------------------------------------------------------------------
// extendMatch flank, grid form (same result as lm_extend_right): bit-parallel 2-mer pairs + Chainer3 on the (q, t) grid.
// * pairs: B[b] has bit t set when target flank base t is b, so the target positions matching the query 2-mer (a,b) at q
//   are rows[q] = B[a] & (B[b] >> 1): no inner loop over the target flank (<= 128 bases; longer flanks use the caller's
//   plain path);
// * chaining: in lm_run_chain3 every predecessor that can be accepted has 1 <= dq <= 10 (max_distance), 0 <= dt <= 10
//   and |dq - dt| <= 5 (max_gap), and the scan limit (band_base / band_count) only ever stops at anchors more than 12
//   query positions back. So instead of walking ~70 earlier anchors per anchor, the ten previous rows are read as bit
//   masks and only the set bits inside the allowed window are visited (~7). Ties: the earliest anchor among equal best
//   scores, and a predecessor beats "start a new chain" on equality - what the descending scan with `>=` produces.
// Scratch is strided (element j at [j * stride]) so a wavefront can interleave its 64 work items.
#define LM_EXT_ROWS 132 /* query-flank rows: ext_len2 (50) + 80 = 130 at most */
struct LmM128 {
    uint64_t lo, hi;
};
LM_HD LmM128 lm_m128_range(int a, int b) { // bits [a, b], 0 <= a <= b <= 127
    LmM128 r;
    const uint64_t lo_from = a >= 64 ? 0ull : (~0ull << a), hi_from = a >= 64 ? (~0ull << (a - 64)) : ~0ull;
    const uint64_t lo_to = b >= 63 ? ~0ull : ((1ull << (b + 1)) - 1ull);
    const uint64_t hi_to = b < 64 ? 0ull : (b >= 127 ? ~0ull : ((1ull << (b - 63)) - 1ull));
    r.lo = lo_from & lo_to;
    r.hi = hi_from & hi_to;
    return r;
}
LM_HD int lm_m128_count_below(const LmM128 &m, int t) { // set bits at positions < t
    if (t <= 0) return 0;
    if (t < 64) return __builtin_popcountll(m.lo & ((1ull << t) - 1ull));
    if (t == 64) return __builtin_popcountll(m.lo);
    return __builtin_popcountll(m.lo) + __builtin_popcountll(m.hi & (t >= 128 ? ~0ull : ((1ull << (t - 64)) - 1ull)));
}
LM_HDN bool lm_chain3_grid(const LmM128 *rows, const uint32_t *rstart, int n1, const uint16_t *subs, int n, int32_t *msi,
                           int stride, int *qend_out, int *tend_out) {
    if (n <= 0) return false;
    int M = 0, Mi = 0, i = 0;
    for (int p = 0; p + 1 < n1; p++) {
        const LmM128 cur = rows[(int64_t)p * stride];
        LmM128 prev[10]; // the (up to) ten previous rows, loaded together: independent loads, one latency
        uint32_t pstart[10];
#pragma unroll
        for (int dq = 1; dq <= 10; dq++) {
            const int q2 = p - dq;
            if (q2 >= 0) {
                prev[dq - 1] = rows[(int64_t)q2 * stride];
                pstart[dq - 1] = rstart[(int64_t)q2 * stride];
            } else {
                prev[dq - 1].lo = prev[dq - 1].hi = 0;
                pstart[dq - 1] = 0;
            }
        }
        for (int half = 0; half < 2; half++) {
            uint64_t bits = half ? cur.hi : cur.lo;
            while (bits) {
                const int at = (half << 6) + __builtin_ctzll(bits);
                bits &= bits - 1;
                const int aq = p;
                const int base = 2 - (aq > at ? aq : at) - (aq > at ? aq - at : at - aq);
                int best = -2147483647, bj = -1;
#pragma unroll
                for (int dq = 10; dq >= 1; dq--) { // far rows first = ascending anchor index
                    int tlo = at - (dq + 5 < 10 ? dq + 5 : 10), thi = at - (dq > 5 ? dq - 5 : 0);
                    if (thi < 0) continue;
                    if (tlo < 0) tlo = 0;
                    const LmM128 pr = prev[dq - 1];
                    const LmM128 rg = lm_m128_range(tlo, thi);
                    uint64_t c0 = pr.lo & rg.lo, c1 = pr.hi & rg.hi;
                    while (c0 | c1) {
                        int bt;
                        if (c0) {
                            bt = __builtin_ctzll(c0);
                            c0 &= c0 - 1;
                        } else {
                            bt = 64 + __builtin_ctzll(c1);
                            c1 &= c1 - 1;
                        }
                        const int j = (int)pstart[dq - 1] + lm_m128_count_below(pr, bt);
                        const int dt = at - bt;
                        const int d = dq > dt ? dq : dt;
                        const int g = dq > dt ? dq - dt : dt - dq;
                        const int sc = (msi[(int64_t)j * stride] >> 16) + 2 - d - g;
                        if (sc > best) {
                            best = sc;
                            bj = j;
                        }
                    }
                }
                int m = base, mj = i;
                if (bj >= 0 && best >= base) {
                    m = best;
                    mj = bj;
                }
                msi[(int64_t)i * stride] = (int32_t)(((uint32_t)m << 16) | (uint32_t)(mj & 0xffff));
                if (i > 0 && m > M) {
                    M = m;
                    Mi = i;
                }
                i++;
            }
        }
    }
    if (M < 1) return false;
    int n_matched = 0, n_abq = 0, n_abt = 0;
    int k = Mi;
    int qb = 0, qe = 0, tb = 0, te = 0, begin_of_next = 0;
    bool first_anchor = true;
    while (true) {
        const int j = (int)((uint32_t)msi[(int64_t)k * stride] & 0xffffu);
        const uint32_t sb = subs[(int64_t)k * stride];
        const int sq = (int)(sb & 255u), st = (int)(sb >> 8);
        if (first_anchor) {
            first_anchor = false;
            qe = sq + 1;
            te = st + 1;
            qb = sq;
            tb = st;
            n_matched += 2;
        } else {
            qb = sq;
            tb = st;
            if (sq + 1 >= begin_of_next)
                n_matched += begin_of_next - sq;
            else
                n_matched += 2;
        }
        begin_of_next = sq;
        if (k == j) {
            n_abq += qe - qb + 1;
            if (n_abq < 2) return false;
            n_abt += te - tb + 1;
            const double pident = (double)n_matched / (double)(n_abq > n_abt ? n_abq : n_abt) * 100;
            if (pident < 15) return false;
            *qend_out = qe;
            *tend_out = te;
            return true;
        }
        k = j;
    }
}
// Chainer3 over compressed anchors in list form (fallback when the flanks do not fit the grid)
LM_HDN bool lm_chain3_list(const uint16_t *subs, int n, int32_t *msi, int stride, int *qend_out, int *tend_out) {
    if (n <= 0) return false;
    int M = 0, Mi = 0;
    for (int i = 0; i < n; i++) {
        const uint32_t a = subs[(int64_t)i * stride];
        const int aq = (int)(a & 255u), at = (int)(a >> 8);
        int m = 2 - (aq > at ? aq : at) - (aq > at ? aq - at : at - aq);
        int mj = i;
        int bcount = 0;
        for (int j = i - 1; j >= 0; j--) {
            const uint32_t b = subs[(int64_t)j * stride];
            const int bq = (int)(b & 255u), bt = (int)(b >> 8);
            if (bq == aq || bt > at) continue;
            bcount++;
            const int bbase = aq - bq - 2;
            if (!(bbase <= 10 || bcount <= 20)) break;
            int dq = aq - bq, dt = at - bt;
            if (dq < 0) dq = -dq;
            if (dt < 0) dt = -dt;
            const int d = dq > dt ? dq : dt;
            if (d > 10) continue;
            const int g = dq > dt ? dq - dt : dt - dq;
            if (g > 5) continue;
            const int sc = (msi[(int64_t)j * stride] >> 16) + 2 - d - g;
            if (sc >= m) {
                m = sc;
                mj = j;
            }
        }
        msi[(int64_t)i * stride] = (int32_t)(((uint32_t)m << 16) | (uint32_t)(mj & 0xffff));
        if (i > 0 && m > M) {
            M = m;
            Mi = i;
        }
    }
    if (M < 1) return false;
    int n_matched = 0, n_abq = 0, n_abt = 0;
    int i = Mi;
    int qb = 0, qe = 0, tb = 0, te = 0, begin_of_next = 0;
    bool first_anchor = true;
    while (true) {
        const int j = (int)((uint32_t)msi[(int64_t)i * stride] & 0xffffu);
        const uint32_t sb = subs[(int64_t)i * stride];
        const int sq = (int)(sb & 255u), st = (int)(sb >> 8);
        if (first_anchor) {
            first_anchor = false;
            qe = sq + 1;
            te = st + 1;
            qb = sq;
            tb = st;
            n_matched += 2;
        } else {
            qb = sq;
            tb = st;
            if (sq + 1 >= begin_of_next)
                n_matched += begin_of_next - sq;
            else
                n_matched += 2;
        }
        begin_of_next = sq;
        if (i == j) {
            n_abq += qe - qb + 1;
            if (n_abq < 2) return false;
            n_abt += te - tb + 1;
            const double pident = (double)n_matched / (double)(n_abq > n_abt ? n_abq : n_abt) * 100;
            if (pident < 15) return false;
            *qend_out = qe;
            *tend_out = te;
            return true;
        }
        i = j;
    }
}
// subs: cap entries (q | t << 8), msi: cap entries, rows / rstart: LM_EXT_ROWS entries, all with `stride`
LM_HDN void lm_extend_flank_grid(const uint8_t *s1, int n1, const uint8_t *s2, int n2, bool rev, uint16_t *subs,
                                 int32_t *msi, int cap, LmM128 *rows, uint32_t *rstart, int stride, int *o1, int *o2) {
    *o1 = 0;
    *o2 = 0;
    if (n1 < 2 || n2 < 2 || n1 > 255 || n2 > 255) return;
    int n = 0;
    const bool grid = n2 <= 128 && n1 <= LM_EXT_ROWS;
    if (!grid) { // plain double loop (only reachable with the +80 extension of > 1 Mb alignments)
        for (int p = 0; p + 1 < n1; p++) {
            const uint32_t km1 = (lm_base2bit(lm_flank_base(s1, n1, p, rev)) << 2) | lm_base2bit(lm_flank_base(s1, n1, p + 1, rev));
            for (int t = 0; t + 1 < n2; t++) {
                const uint32_t km2 =
                    (lm_base2bit(lm_flank_base(s2, n2, t, rev)) << 2) | lm_base2bit(lm_flank_base(s2, n2, t + 1, rev));
                if (km1 == km2) {
                    if (n >= cap) return;
                    subs[(int64_t)(n++) * stride] = (uint16_t)(p | (t << 8));
                }
            }
        }
    } else {
        LmM128 B0 = {0, 0}, B1 = {0, 0}, B2 = {0, 0}, B3 = {0, 0};
        for (int t = 0; t < n2; t++) {
            const uint32_t c = lm_base2bit(lm_flank_base(s2, n2, t, rev));
            const uint64_t bl = t < 64 ? 1ull << t : 0ull, bh = t >= 64 ? 1ull << (t - 64) : 0ull;
            if (c == 0) { B0.lo |= bl; B0.hi |= bh; }
            else if (c == 1) { B1.lo |= bl; B1.hi |= bh; }
            else if (c == 2) { B2.lo |= bl; B2.hi |= bh; }
            else { B3.lo |= bl; B3.hi |= bh; }
        }
        const LmM128 V = lm_m128_range(0, n2 - 2); // valid 2-mer starts
        uint32_t ca = lm_base2bit(lm_flank_base(s1, n1, 0, rev));
        for (int p = 0; p + 1 < n1; p++) {
            const uint32_t cb = lm_base2bit(lm_flank_base(s1, n1, p + 1, rev));
            const LmM128 A = ca == 0 ? B0 : ca == 1 ? B1 : ca == 2 ? B2 : B3;
            const LmM128 Bn = cb == 0 ? B0 : cb == 1 ? B1 : cb == 2 ? B2 : B3;
            uint64_t mlo = A.lo & ((Bn.lo >> 1) | (Bn.hi << 63)) & V.lo;
            uint64_t mhi = A.hi & (Bn.hi >> 1) & V.hi;
            LmM128 row = {mlo, mhi};
            rows[(int64_t)p * stride] = row;
            rstart[(int64_t)p * stride] = (uint32_t)n;
            while (mlo) {
                const int t = __builtin_ctzll(mlo);
                mlo &= mlo - 1;
                if (n >= cap) return;
                subs[(int64_t)(n++) * stride] = (uint16_t)(p | (t << 8));
            }
            while (mhi) {
                const int t = 64 + __builtin_ctzll(mhi);
                mhi &= mhi - 1;
                if (n >= cap) return;
                subs[(int64_t)(n++) * stride] = (uint16_t)(p | (t << 8));
            }
            ca = cb;
        }
    }
    if (n == 0) return;
    int qe, te;
    if (grid ? lm_chain3_grid(rows, rstart, n1, subs, n, msi, stride, &qe, &te)
             : lm_chain3_list(subs, n, msi, stride, &qe, &te)) {
        *o1 = qe + 1;
        *o2 = te + 1;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// tree.Search(key, p) over a sorted key array (tree/tree.go:441-527): the range of entries the radix tree would
// return, INCLUDING the partial-prefix quirk at :496-500 (uint8 wrap of n.k-atleast turns the test into
// "bases [d,p) of the key are all A", which returns a subtree whose leaves share fewer than p bases).
// keys[0..n) sorted ascending with duplicates adjacent; k-mers of length K. Returns true and [*lo,*hi) if any.
LM_HD int lm_lower_bound_u64(const uint64_t *a, int lo, int hi, uint64_t x) {
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (a[mid] < x)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}
LM_HD int lm_upper_bound_u64(const uint64_t *a, int lo, int hi, uint64_t x) {
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (a[mid] <= x)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}
// The "no key shares p bases" half of tree.Search: decides whether the partial-prefix quirk returns a subtree.
// lo = insertion point of (key & ~low) in keys[0..n).
LM_HDN bool lm_tree_search_miss(const uint64_t *keys, int n, uint64_t key, int p, int K, int lo, int *lo_out, int *hi_out) {
    const int sh = (K - p) << 1;
    // No key shares p bases. The quirk needs bases [d,p) of `key` to be all A for some node depth d <= L, where
    // L = longest prefix shared with any key; cheap necessary test first: bases [L,p) all A.
    int L = 0;
    if (lo > 0) L = lm_lcp(key, keys[lo - 1], K);
    if (lo < n) {
        int l2 = lm_lcp(key, keys[lo], K);
        if (l2 > L) L = l2;
    }
    if (L < 1) return false; // root has no child for the first base
    {
        // bases [L,p) of key
        uint64_t seg = (key >> sh) & ((p - L) >= 32 ? ~0ull : ((1ull << ((p - L) << 1)) - 1));
        if (seg != 0) return false;
    }
    // Simulate the descent (node = maximal group of keys sharing the edge).
    int rlo = 0, rhi = n, d = 0;
    while (true) {
        if (d >= K) return false;
        // child for base d: keys in [rlo,rhi) sharing first d+1 bases with key
        const int s1 = (K - d - 1) << 1;
        const uint64_t l1 = (1ull << s1) - 1;
        int clo = lm_lower_bound_u64(keys, rlo, rhi, key & ~l1);
        int chi = lm_upper_bound_u64(keys, clo, rhi, key | l1);
        if (clo >= chi) return false;
        int e = keys[clo] == keys[chi - 1] ? K : lm_lcp(keys[clo], keys[chi - 1], K); // node end depth
        int l = lm_lcp(key, keys[clo], K);
        if (l >= e) { // full edge matched
            if (e >= p) { // cannot happen here (normal range was empty) but keep the literal rule
                *lo_out = clo;
                *hi_out = chi;
                return true;
            }
            d = e;
            rlo = clo;
            rhi = chi;
            continue;
        }
        int atleast = p - d, nk = e - d;
        if (nk >= atleast) return false; // would need l >= p, impossible here
        // uint8 wrap: (n.k-atleast)<<1 >= 64 -> rhs 0 ; lhs = bases [d,p) of key
        uint64_t seg = (key >> sh) & (atleast >= 32 ? ~0ull : ((1ull << (atleast << 1)) - 1));
        if (seg == 0) {
            *lo_out = clo;
            *hi_out = chi;
            return true;
        }
        return false;
    }
}


// lm_tree_search_miss with the descent through the levels the bucket table covers read from the table (two loads per level
// instead of two binary searches over the whole array): tab over the leading tab_bits bits, tab[2^tab_bits] = n.
LM_HDN bool lm_tree_search_miss_tab(const uint64_t *keys, int n, uint64_t key, int p, int K, int lo, const uint32_t *tab,
                                    int tab_bits, int *lo_out, int *hi_out) {
    const int sh = (K - p) << 1;
    int L = 0;
    if (lo > 0) L = lm_lcp(key, keys[lo - 1], K);
    if (lo < n) {
        int l2 = lm_lcp(key, keys[lo], K);
        if (l2 > L) L = l2;
    }
    if (L < 1) return false;
    {
        uint64_t seg = (key >> sh) & ((p - L) >= 32 ? ~0ull : ((1ull << ((p - L) << 1)) - 1));
        if (seg != 0) return false;
    }
    int rlo = 0, rhi = n, d = 0;
    while (true) {
        if (d >= K) return false;
        const int s1 = (K - d - 1) << 1;
        const uint64_t l1 = (1ull << s1) - 1;
        int clo, chi;
        if (2 * (d + 1) <= tab_bits) { // the keys sharing d+1 bases with `key` are a run of whole buckets
            const int up = tab_bits - 2 * (d + 1);
            const uint32_t pre = (uint32_t)(key >> ((K << 1) - 2 * (d + 1)));
            clo = (int)tab[pre << up];
            chi = (int)tab[(pre + 1u) << up];
        } else {
            clo = lm_lower_bound_u64(keys, rlo, rhi, key & ~l1);
            chi = lm_upper_bound_u64(keys, clo, rhi, key | l1);
        }
        if (clo >= chi) return false;
        int e = keys[clo] == keys[chi - 1] ? K : lm_lcp(keys[clo], keys[chi - 1], K);
        int l = lm_lcp(key, keys[clo], K);
        if (l >= e) {
            if (e >= p) {
                *lo_out = clo;
                *hi_out = chi;
                return true;
            }
            d = e;
            rlo = clo;
            rhi = chi;
            continue;
        }
        int atleast = p - d, nk = e - d;
        if (nk >= atleast) return false;
        uint64_t seg = (key >> sh) & (atleast >= 32 ? ~0ull : ((1ull << (atleast << 1)) - 1));
        if (seg == 0) {
            *lo_out = clo;
            *hi_out = chi;
            return true;
        }
        return false;
    }
}

LM_HDN bool lm_tree_search_range(const uint64_t *keys, int n, uint64_t key, int p, int K, int *lo_out, int *hi_out) {
    if (n <= 0) return false;
    if (p < 1) p = 1;
    if (p > K) p = K;
    const int sh = (K - p) << 1;
    const uint64_t low = sh >= 64 ? ~0ull : ((1ull << sh) - 1);
    const uint64_t left = key & ~low, right = key | low;
    int lo = lm_lower_bound_u64(keys, 0, n, left);
    int hi = lm_upper_bound_u64(keys, lo, n, right);
    if (lo < hi) {
        *lo_out = lo;
        *hi_out = hi;
        return true;
    }
    // the quirk needs bases [L,p) of the key to be A for some L < p: base p-1 is A or nothing is returned
    if ((key >> sh) & 3ull) return false;
    return lm_tree_search_miss(keys, n, key, p, K, lo, lo_out, hi_out);
}

// Prefix filter of the pseudo-alignment (k_build_cmp_bits / k_pa_filter).  Per query, lm_pa_bits_words(log) words:
//   [0, W)        hashed bitmap of the 11-base prefixes of its k-mers, 2^log bits (~16 per k-mer), W = 2^(log-5)
//   [W, 2W)       hashed bitmap of the 9-base prefixes, 2^log bits            (generic path only: lm_pa_candidate)
//   [2W, 2W + B)  Bloom filter of the 11-base prefixes, two hash functions, 2^blog bits, blog = min(log, 19), B = 2^(blog-5)
//   [.., + 8192)  exact bitmap of the 9-base prefixes (4^9 = 2^18 bits)
// The last two (<= 96 KB) are what a workgroup of k_pa_filter keeps in LDS for the query it works on: a chain window of
// an unrelated genome (most windows of a search against 10^5 genomes: random 17-base seed matches) is then rejected
// position by position without a single global load; the 11-base bitmap in global memory is only asked for positions the
// Bloom filter lets through when it is the more selective one (log > blog: reads above ~16 kb).
// lm_pa_candidate / lm_pa_candidate2 are NECESSARY conditions for lm_tree_search_range(keys, key, p) to return true
// (p >= 11):
//   * some query k-mer shares the key's 11-base prefix (a normal match needs >= p >= 11 common bases), or
//   * the 11-base map misses, so the longest common prefix L is <= 10 and only the partial-prefix rule of tree.Search
//     (tree.go:496-500) can fire: at a node of depth d <= L-1 with the key's bases [d, p) all A.  With a = the number of
//     A's that end the key's first p bases, d >= p - a.  d <= 9 forces the bases [9, p) to be A (a >= p - 9); then either
//     a >= p - 7 (bases [7, p) all A: rare, always a candidate) or d >= 8, so L >= 9 and the 9-base map must hit.
//     lm_pa_candidate2 is sharper for the commonest case a = p - 9 (base 8 is not A, 3 in 4): then d = 9 and L = 10
//     exactly, so one of the three 11-base prefixes that differ from the key's in base 10 (an A) must be present.
#define LM_PFX_BASES 11
#define LM_PFX_BASES2 9
#define LM_PA_BLOOM_LOG_MAX 19
#define LM_PA_MAP9_LOG 18
LM_HD uint32_t lm_pa_filter_slot(uint32_t pfx, int log) {
    return (pfx * 0x9E3779B1u) >> (32 - log);
}
LM_HD int lm_pa_bloom_log(int log) { return log < LM_PA_BLOOM_LOG_MAX ? log : LM_PA_BLOOM_LOG_MAX; }
// (a 24-bit multiply: the 11-base prefix has 22 bits, and v_mul_u32_u24 issues at full rate where v_mul_lo_u32 takes four
// passes - four of them per window position were a fifth of k_pa_filter's vector time)
LM_HD uint32_t lm_mul24(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a, b);
#else
    return (uint32_t)((uint64_t)(a & 0xffffffu) * (uint64_t)(b & 0xffffffu));
#endif
}
LM_HD uint32_t lm_pa_bloom_slot(uint32_t pfx, int which, int blog) {
    return lm_mul24(pfx, which ? 0xB2AE35u : 0xEBCA6Bu) >> (32 - blog);
}
LM_HD uint64_t lm_pa_bits_words(int log) {
    return 2 * ((uint64_t)1 << (log - 5)) + ((uint64_t)1 << (lm_pa_bloom_log(log) - 5)) + ((uint64_t)1 << (LM_PA_MAP9_LOG - 5));
}
LM_HD uint64_t lm_pa_bloom_word0(int log) { return 2 * ((uint64_t)1 << (log - 5)); }
LM_HD uint64_t lm_pa_map9_word0(int log) { return lm_pa_bloom_word0(log) + ((uint64_t)1 << (lm_pa_bloom_log(log) - 5)); }
// every bit a k-mer of the query sets (`or_bit(word index, mask)`: atomicOr on the device)
template <typename OrBit> LM_HD void lm_pa_filter_set(uint64_t key, int K, int log, OrBit or_bit) {
    const uint32_t p11 = (uint32_t)(key >> ((K - LM_PFX_BASES) << 1)), p9 = (uint32_t)(key >> ((K - LM_PFX_BASES2) << 1));
    const int blog = lm_pa_bloom_log(log);
    const uint32_t h = lm_pa_filter_slot(p11, log), h2 = lm_pa_filter_slot(p9, log);
    or_bit((uint64_t)(h >> 5), 1u << (h & 31));
    or_bit(((uint64_t)1 << (log - 5)) + (h2 >> 5), 1u << (h2 & 31));
    const uint32_t a = lm_pa_bloom_slot(p11, 0, blog), b = lm_pa_bloom_slot(p11, 1, blog);
    or_bit(lm_pa_bloom_word0(log) + (a >> 5), 1u << (a & 31));
    or_bit(lm_pa_bloom_word0(log) + (b >> 5), 1u << (b & 31));
    or_bit(lm_pa_map9_word0(log) + (p9 >> 5), 1u << (p9 & 31));
}
LM_HD bool lm_pa_candidate(const uint32_t *bits, int log, uint64_t key, int p, int K) {
    const uint32_t h = lm_pa_filter_slot((uint32_t)(key >> ((K - LM_PFX_BASES) << 1)), log);
    if ((bits[h >> 5] >> (h & 31)) & 1u) return true;
    const uint64_t first_p = key >> ((K - p) << 1);
    if (first_p & ((1ull << ((p - 9) << 1)) - 1ull)) return false; // bases [9, p) not all A
    if ((first_p & ((1ull << ((p - 7) << 1)) - 1ull)) == 0) return true; // bases [7, p) all A
    const uint32_t h2 = lm_pa_filter_slot((uint32_t)(key >> ((K - LM_PFX_BASES2) << 1)), log);
    const uint32_t *bits2 = bits + ((uint64_t)1 << (log - 5));
    return ((bits2[h2 >> 5] >> (h2 & 31)) & 1u) != 0;
}
// the test k_pa_filter runs per (window position, strand): `f` = the first p bases of the k-mer (11 <= p <= 15),
// `bloom` / `map9` = the query's Bloom filter and exact 9-base map (LDS copies on the device), `bits11` = its hashed
// 11-base bitmap in global memory
LM_HD bool lm_pa_candidate2(const uint32_t *bloom, int blog, const uint32_t *map9, const uint32_t *bits11, int log,
                            uint32_t f, int p) {
    const uint32_t p11 = f >> ((p - LM_PFX_BASES) << 1);
    const uint32_t a = lm_pa_bloom_slot(p11, 0, blog), b = lm_pa_bloom_slot(p11, 1, blog);
    bool hit = (((bloom[a >> 5] >> (a & 31)) & (bloom[b >> 5] >> (b & 31))) & 1u) != 0;
    if (hit && log > blog) {
        const uint32_t h = lm_pa_filter_slot(p11, log);
        hit = ((bits11[h >> 5] >> (h & 31)) & 1u) != 0;
    }
    if (hit) return true;
    if (f & ((1u << ((p - 9) << 1)) - 1u)) return false; // bases [9, p) not all A
    if ((f & ((1u << ((p - 7) << 1)) - 1u)) == 0) return true; // bases [7, p) all A
    const uint32_t p9 = f >> ((p - LM_PFX_BASES2) << 1);
    const bool m9 = ((map9[p9 >> 5] >> (p9 & 31)) & 1u) != 0; // (either way some k-mer shares the key's first 9 bases: the exact map)
    if (!m9) return false;
    if (p9 & 3u) {
        // base 8 is not A: a = p - 9, so d = 9 and L = 10 exactly: some k-mer shares the key's 10 bases and differs at
        // base 10 (an A in the key): one of the three sibling 11-base prefixes is in the set
        for (uint32_t sib = 1; sib < 4; sib++) {
            const uint32_t x = p11 | sib, sa = lm_pa_bloom_slot(x, 0, blog), sb = lm_pa_bloom_slot(x, 1, blog);
            bool h = (((bloom[sa >> 5] >> (sa & 31)) & (bloom[sb >> 5] >> (sb & 31))) & 1u) != 0;
            if (h && log > blog) {
                const uint32_t hh = lm_pa_filter_slot(x, log);
                h = ((bits11[hh >> 5] >> (hh & 31)) & 1u) != 0;
            }
            if (h) return true;
        }
        return false;
    }
    return true; // base 8 is an A: the 9-base map was the test
}

// Same, with the two binary searches narrowed by a bucket table over the leading `tab_bits/2` bases:
// tab[b] = first index whose leading bits are >= b, tab[nbuckets] = n.  Requires 2*p >= tab_bits.
LM_HD bool lm_tree_search_range_tab(const uint64_t *keys, int n, uint64_t key, int p, int K, const uint32_t *tab,
                                    int tab_bits, int *lo_out, int *hi_out) {
    if (n <= 0) return false;
    if (p > K) p = K;
    const int sh = (K - p) << 1;
    const uint64_t low = sh >= 64 ? ~0ull : ((1ull << sh) - 1);
    const uint64_t left = key & ~low, right = key | low;
    const uint32_t b = (uint32_t)(key >> ((K << 1) - tab_bits));
    int lo = lm_lower_bound_u64(keys, (int)tab[b], (int)tab[b + 1], left);
    int hi = lm_upper_bound_u64(keys, lo, (int)tab[b + 1], right);
    if (lo < hi) {
        *lo_out = lo;
        *hi_out = hi;
        return true;
    }
    // the quirk needs bases [L,p) of the key to be A for some L < p: base p-1 is A or nothing is returned
    if ((key >> sh) & 3ull) return false;
    return lm_tree_search_miss(keys, n, key, p, K, lo, lo_out, hi_out);
}

// The form k_pa_search uses: only the lower bound is searched.  Returns 1 when keys share >= p bases with `key`: the
// matches are keys[*lo_out], keys[*lo_out + 1], ... while they stay <= *right_out (the caller enumerates them anyway, so the
// upper-bound search would be wasted loads); 2 when the partial-prefix quirk returns the subtree [*lo_out, *hi_out); 0 for
// no result.  Same results as lm_tree_search_range (checked on the CPU against the reference radix tree).
LM_HD int lm_tree_search_first_tab(const uint64_t *keys, int n, uint64_t key, int p, int K, const uint32_t *tab, int tab_bits,
                                   int *lo_out, int *hi_out, uint64_t *right_out) {
    if (n <= 0) return 0;
    if (p > K) p = K;
    const int sh = (K - p) << 1;
    const uint64_t low = sh >= 64 ? ~0ull : ((1ull << sh) - 1);
    const uint64_t left = key & ~low, right = key | low;
    const uint32_t b = (uint32_t)(key >> ((K << 1) - tab_bits));
    const int bend = (int)tab[b + 1];
    const int lo = lm_lower_bound_u64(keys, (int)tab[b], bend, left);
    if (lo < bend && keys[lo] <= right) { // keys sharing p bases lie in one bucket (2p >= tab_bits)
        *lo_out = lo;
        *hi_out = bend;
        *right_out = right;
        return 1;
    }
    if ((key >> sh) & 3ull) return 0; // the quirk needs base p-1 of the key to be A
    *right_out = ~0ull;
    return lm_tree_search_miss_tab(keys, n, key, p, K, lo, tab, tab_bits, lo_out, hi_out) ? 2 : 0;
}

// ---------------------------------------------------------------------------------------------------------------
// WFA, gap-affine (x=4,o=6,e=2), end-to-end, wf-adaptive(10,50,1), WFA2 backtrace priority — one work item.
// Restates github.com/shenwei356/wfa v0.5.0 as used at lib-index-search.go:1910-1911,2261,2528 (not in the reference
// tree): pattern = query (v), text = target (h), k = h - v, offset = h; 'I' consumes target, 'D' consumes query;
// backtrace priority on equal offsets mismatch > D-ext > D-open > I-ext > I-open (WFA2 piggy-back codes); reads
// outside a stored wavefront's [lo,hi] are NULL; cut-off applied after every extension.
// Memory: `hdr` holds per score 3x(lo,hi,base) int32 = 9*max_score ints; `arena` holds offsets.
struct LmWfaOut {
    int32_t status; // 0 ok, 1 arena/score overflow (retry with more memory), 2 no alignment
    int32_t score;
    int32_t nops;   // run-length ops written (forward order) into ops[]
    int32_t qbegin, qend, tbegin, tend; // 1-based, first..last M
    uint32_t align_len, matches, gaps, gap_regions;
};

#define LM_WF_M 0
#define LM_WF_I 1
#define LM_WF_D 2

LM_HD int32_t lm_wf_get(const int32_t *hdr, const int32_t *arena, int comp, int s, int k) {
    if (s < 0) return LM_NULL_OFF;
    const int32_t *h = hdr + (s * 3 + comp) * 3;
    if (k < h[0] || k > h[1]) return LM_NULL_OFF;
    return arena[h[2] + (k - h[0])];
}

// trim invalid ends of component `comp` at score s (WFA2 wavefront_compute_trim_ends)
LM_HD void lm_wf_trim(int32_t *hdr, const int32_t *arena, int comp, int s, int plen, int tlen, int alo) {
    int32_t *h = hdr + (s * 3 + comp) * 3;
    int lo = h[0], hi = h[1], base = h[2];
    int k;
    for (k = hi; k >= lo; --k) {
        int32_t off = arena[base + (k - alo)];
        uint32_t hh = (uint32_t)off, vv = (uint32_t)(off - k);
        if (hh <= (uint32_t)tlen && vv <= (uint32_t)plen) break;
    }
    hi = k;
    for (k = lo; k <= hi; ++k) {
        int32_t off = arena[base + (k - alo)];
        uint32_t hh = (uint32_t)off, vv = (uint32_t)(off - k);
        if (hh <= (uint32_t)tlen && vv <= (uint32_t)plen) break;
    }
    lo = k;
    // keep `base` addressing relative to the new lo
    h[2] = base + (lo - alo);
    h[0] = lo;
    h[1] = hi;
}

// Backtrace (WFA2 wavefront_backtrace_affine) + statistics over the stored wavefronts; `s` = final score.
// ops: output buffer of packed (op<<32|n) runs, capacity ops_cap; built backwards from the end of the buffer and then
// moved to the front.
LM_HDN void lm_wfa_backtrace(const int32_t *hdr, const int32_t *arena, int s, int plen, int tlen, uint64_t *ops,
                             int ops_cap, LmWfaOut *out) {
    const int X = 4, OE = 8, E = 2;
    const int ak = tlen - plen;
    out->status = 0;
    out->nops = 0;
    out->qbegin = out->qend = out->tbegin = out->tend = 0;
    out->align_len = out->matches = out->gaps = out->gap_regions = 0;
    out->score = s;
    int wp = ops_cap; // next write slot is wp-1
    char cur_op = 0;
    uint32_t cur_n = 0;
    bool overflow = false;
#define LM_PUSH(OP, N)                                                     \
    do {                                                                   \
        int _n = (N);                                                      \
        if (_n > 0) {                                                      \
            if (cur_op == (OP)) {                                          \
                cur_n += (uint32_t)_n;                                     \
            } else {                                                       \
                if (cur_n) {                                               \
                    if (wp <= 0) overflow = true;                          \
                    else ops[--wp] = ((uint64_t)(uint8_t)cur_op << 32) | cur_n; \
                }                                                          \
                cur_op = (OP);                                             \
                cur_n = (uint32_t)_n;                                      \
            }                                                              \
        }                                                                  \
    } while (0)
    int score = s, k = ak;
    int32_t offset = tlen;
    int v = offset - k, h = offset;
    int matrix = 0;
    const int64_t NEG = (int64_t)LM_NULL_OFF * 16;
    while (v > 0 && h > 0 && score > 0) {
        int s_mis = score - X, s_open = score - OE, s_ext = score - E;
        int64_t c_mis = NEG, c_io = NEG, c_ie = NEG, c_do = NEG, c_de = NEG;
        if (matrix == 0) {
            int64_t o;
            o = (int64_t)lm_wf_get(hdr, arena, LM_WF_M, s_mis, k) + 1;
            if (o >= 0) c_mis = (o << 4) | 9;
            o = (int64_t)lm_wf_get(hdr, arena, LM_WF_M, s_open, k - 1) + 1;
            if (o >= 0) c_io = (o << 4) | 1;
            o = (int64_t)lm_wf_get(hdr, arena, LM_WF_M, s_open, k + 1);
            if (o >= 0) c_do = (o << 4) | 3;
            o = (int64_t)lm_wf_get(hdr, arena, LM_WF_I, s_ext, k - 1) + 1;
            if (o >= 0) c_ie = (o << 4) | 2;
            o = (int64_t)lm_wf_get(hdr, arena, LM_WF_D, s_ext, k + 1);
            if (o >= 0) c_de = (o << 4) | 4;
        } else if (matrix == 1) {
            int64_t o;
            o = (int64_t)lm_wf_get(hdr, arena, LM_WF_M, s_open, k - 1) + 1;
            if (o >= 0) c_io = (o << 4) | 1;
            o = (int64_t)lm_wf_get(hdr, arena, LM_WF_I, s_ext, k - 1) + 1;
            if (o >= 0) c_ie = (o << 4) | 2;
        } else {
            int64_t o;
            o = (int64_t)lm_wf_get(hdr, arena, LM_WF_M, s_open, k + 1);
            if (o >= 0) c_do = (o << 4) | 3;
            o = (int64_t)lm_wf_get(hdr, arena, LM_WF_D, s_ext, k + 1);
            if (o >= 0) c_de = (o << 4) | 4;
        }
        int64_t mx = c_mis;
        if (c_io > mx) mx = c_io;
        if (c_ie > mx) mx = c_ie;
        if (c_do > mx) mx = c_do;
        if (c_de > mx) mx = c_de;
        if (mx < 0) break;
        if (matrix == 0) {
            int32_t max_off = (int32_t)(mx >> 4);
            LM_PUSH('M', offset - max_off);
            offset = max_off;
            v = offset - k;
            h = offset;
            if (v <= 0 || h <= 0) break;
        }
        int bt = (int)(mx & 15);
        if (bt == 9) {
            score = s_mis; matrix = 0; LM_PUSH('X', 1); --offset;
        } else if (bt == 1) {
            score = s_open; matrix = 0; LM_PUSH('I', 1); --k; --offset;
        } else if (bt == 2) {
            score = s_ext; matrix = 1; LM_PUSH('I', 1); --k; --offset;
        } else if (bt == 3) {
            score = s_open; matrix = 0; LM_PUSH('D', 1); ++k;
        } else {
            score = s_ext; matrix = 2; LM_PUSH('D', 1); ++k;
        }
        v = offset - k;
        h = offset;
    }
    if (v > 0 && h > 0) {
        int nm = v < h ? v : h;
        LM_PUSH('M', nm);
        v -= nm;
        h -= nm;
    }
    if (v > 0) LM_PUSH('D', v);
    if (h > 0) LM_PUSH('I', h);
    if (cur_n) {
        if (wp <= 0) overflow = true;
        else ops[--wp] = ((uint64_t)(uint8_t)cur_op << 32) | cur_n;
    }
#undef LM_PUSH
    if (overflow) {
        out->status = 1;
        return;
    }
    int nops = ops_cap - wp;
    for (int i = 0; i < nops; i++) ops[i] = ops[wp + i];
    out->nops = nops;
    int first = -1, last = -1;
    for (int i = 0; i < nops; i++)
        if ((ops[i] >> 32) == 'M') {
            if (first < 0) first = i;
            last = i;
        }
    if (first < 0) {
        out->status = 2;
        return;
    }
    int qpos = 0, tpos = 0;
    for (int i = 0; i < nops; i++) {
        char op = (char)(ops[i] >> 32);
        int nn = (int)(ops[i] & 0xffffffffu);
        if (i == first) {
            out->qbegin = qpos + 1;
            out->tbegin = tpos + 1;
        }
        if (op == 'M' || op == 'X') {
            qpos += nn;
            tpos += nn;
        } else if (op == 'I') {
            tpos += nn;
        } else {
            qpos += nn;
        }
        if (i >= first && i <= last) {
            out->align_len += (uint32_t)nn;
            if (op == 'M') out->matches += (uint32_t)nn;
            if (op == 'I' || op == 'D') {
                out->gaps += (uint32_t)nn;
                out->gap_regions++;
            }
        }
        if (i == last) {
            out->qend = qpos;
            out->tend = tpos;
        }
    }
}

LM_HDN void lm_wfa_align(const uint8_t *q, int plen, const uint8_t *t, int tlen, int32_t *hdr, int max_score,
                         int32_t *arena, int64_t arena_cap, uint64_t *ops, int ops_cap, LmWfaOut *out) {
    const int X = 4, OE = 8, E = 2;
    out->status = 0;
    out->nops = 0;
    out->qbegin = out->qend = out->tbegin = out->tend = 0;
    out->align_len = out->matches = out->gaps = out->gap_regions = 0;
    int64_t used = 0;
    if (max_score < 1 || arena_cap < 1) {
        out->status = 1;
        return;
    }
    // score 0
    for (int c = 0; c < 3; c++) {
        hdr[c * 3 + 0] = 1;
        hdr[c * 3 + 1] = -1;
        hdr[c * 3 + 2] = 0;
    }
    hdr[0] = 0;
    hdr[1] = 0;
    hdr[2] = 0;
    arena[used++] = 0;
    const int ak = tlen - plen;
    int s = 0;
    while (true) {
        int32_t *hm = hdr + (s * 3 + LM_WF_M) * 3;
        if (hm[0] <= hm[1]) {
            // extend
            for (int k = hm[0]; k <= hm[1]; k++) {
                int32_t off = arena[hm[2] + (k - hm[0])];
                if (off < 0) continue;
                int v = off - k, h = off;
                while (v < plen && h < tlen && q[v] == t[h]) {
                    v++;
                    h++;
                }
                arena[hm[2] + (k - hm[0])] = h;
            }
            if (hm[0] <= ak && ak <= hm[1] && arena[hm[2] + (ak - hm[0])] >= tlen) break;
            // wf-adaptive cut-off
            {
                int lo = hm[0], hi = hm[1];
                if (hi - lo + 1 >= 10) {
                    int min_d = 2147483647;
                    for (int k = lo; k <= hi; k++) {
                        int32_t off = arena[hm[2] + (k - lo)];
                        int d;
                        if (off < 0) {
                            d = 1073741824;
                        } else {
                            int lv = plen - (off - k), lh = tlen - off;
                            d = lv > lh ? lv : lh;
                        }
                        if (d < min_d) min_d = d;
                    }
                    int nlo = lo, nhi = hi;
                    int top = ak < hi ? ak : hi;
                    for (int k = lo; k < top; ++k) {
                        int32_t off = arena[hm[2] + (k - lo)];
                        int d;
                        if (off < 0) {
                            d = 1073741824;
                        } else {
                            int lv = plen - (off - k), lh = tlen - off;
                            d = lv > lh ? lv : lh;
                        }
                        if (d - min_d <= 50) break;
                        ++nlo;
                    }
                    int bottom = ak > nlo ? ak : nlo;
                    for (int k = hi; k > bottom; --k) {
                        int32_t off = arena[hm[2] + (k - lo)];
                        int d;
                        if (off < 0) {
                            d = 1073741824;
                        } else {
                            int lv = plen - (off - k), lh = tlen - off;
                            d = lv > lh ? lv : lh;
                        }
                        if (d - min_d <= 50) break;
                        --nhi;
                    }
                    hm[2] += nlo - lo;
                    hm[0] = nlo;
                    hm[1] = nhi;
                }
                // equate I and D to M
                for (int c = LM_WF_I; c <= LM_WF_D; c++) {
                    int32_t *hc = hdr + (s * 3 + c) * 3;
                    if (hc[0] > hc[1]) continue;
                    if (hm[0] > hc[0]) {
                        hc[2] += hm[0] - hc[0];
                        hc[0] = hm[0];
                    }
                    if (hm[1] < hc[1]) hc[1] = hm[1];
                }
            }
        }
        s++;
        if (s >= max_score) {
            out->status = 1;
            return;
        }
        // compute score s
        int32_t *ho = hdr + s * 9;
        for (int c = 0; c < 3; c++) {
            ho[c * 3 + 0] = 1;
            ho[c * 3 + 1] = -1;
            ho[c * 3 + 2] = 0;
        }
        int lo = 2147483647, hi = -2147483647;
        bool any = false;
        if (s - X >= 0) {
            const int32_t *h = hdr + ((s - X) * 3 + LM_WF_M) * 3;
            if (h[0] <= h[1]) {
                any = true;
                if (h[0] < lo) lo = h[0];
                if (h[1] > hi) hi = h[1];
            }
        }
        if (s - OE >= 0) {
            const int32_t *h = hdr + ((s - OE) * 3 + LM_WF_M) * 3;
            if (h[0] <= h[1]) {
                any = true;
                if (h[0] - 1 < lo) lo = h[0] - 1;
                if (h[1] + 1 > hi) hi = h[1] + 1;
            }
        }
        if (s - E >= 0) {
            const int32_t *h = hdr + ((s - E) * 3 + LM_WF_I) * 3;
            if (h[0] <= h[1]) {
                any = true;
                if (h[0] + 1 < lo) lo = h[0] + 1;
                if (h[1] + 1 > hi) hi = h[1] + 1;
            }
            h = hdr + ((s - E) * 3 + LM_WF_D) * 3;
            if (h[0] <= h[1]) {
                any = true;
                if (h[0] - 1 < lo) lo = h[0] - 1;
                if (h[1] - 1 > hi) hi = h[1] - 1;
            }
        }
        if (!any || lo > hi) continue;
        int w = hi - lo + 1;
        if (used + 3ll * w > arena_cap) {
            out->status = 1;
            return;
        }
        int bm = (int)used, bi = (int)(used + w), bd = (int)(used + 2ll * w);
        used += 3ll * w;
        for (int k = lo; k <= hi; k++) {
            int32_t a = lm_wf_get(hdr, arena, LM_WF_M, s - OE, k - 1), b = lm_wf_get(hdr, arena, LM_WF_I, s - E, k - 1);
            int32_t ins = (a > b ? a : b) + 1;
            a = lm_wf_get(hdr, arena, LM_WF_M, s - OE, k + 1);
            b = lm_wf_get(hdr, arena, LM_WF_D, s - E, k + 1);
            int32_t del = a > b ? a : b;
            int32_t mis = lm_wf_get(hdr, arena, LM_WF_M, s - X, k) + 1;
            int32_t mx = mis > ins ? mis : ins;
            if (del > mx) mx = del;
            uint32_t hh = (uint32_t)mx, vv = (uint32_t)(mx - k);
            if (hh > (uint32_t)tlen) mx = LM_NULL_OFF;
            if (vv > (uint32_t)plen) mx = LM_NULL_OFF;
            arena[bi + (k - lo)] = ins;
            arena[bd + (k - lo)] = del;
            arena[bm + (k - lo)] = mx;
        }
        ho[0] = lo; ho[1] = hi; ho[2] = bm;
        ho[3] = lo; ho[4] = hi; ho[5] = bi;
        ho[6] = lo; ho[7] = hi; ho[8] = bd;
        lm_wf_trim(hdr, arena, LM_WF_M, s, plen, tlen, lo);
        lm_wf_trim(hdr, arena, LM_WF_I, s, plen, tlen, lo);
        lm_wf_trim(hdr, arena, LM_WF_D, s, plen, tlen, lo);
    }
    lm_wfa_backtrace(hdr, arena, s, plen, tlen, ops, ops_cap, out);
}

