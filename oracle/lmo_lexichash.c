/* CPU ORACLE (test infrastructure only) — LexicHash masking.
 *
 * The reference delegates this to github.com/shenwei356/lexichash v0.5.5 (go.mod:24), which is NOT in
 * /root/reference.  Restated from the published LexicHash algorithm and from the reference's call sites:
 *   NewFromFile / IndexMasks(p) / IndexMasksWithDistinctPrefixes(p+1)   lib-index-search.go:430-478
 *   MaskKnownDistinctPrefixes(s, skipRegions, checkShorterPrefix)        lib-index-search.go:1212-1216,
 *                                                                        lib-index-build.go:1028,1198
 *   MaskKmer(kmer)                                                       lib-index-search.go:1328
 * Semantics assumed: for every mask i, the captured k-mer is argmin over ALL k-mers of both
 * strands of (mask_i XOR kmer); every occurrence is reported as (pos<<1|strand), pos = forward-strand window start,
 * in ascending order.  The "known prefixes" variants are accelerations of the same argmin: a k-mer can only be the
 * argmin of a mask sharing its p-base prefix if any k-mer shares that prefix; masks whose p-prefix never occurs are
 * resolved over all k-mers when checkShorterPrefix is true and capture nothing (kmer 0, no locs) when it is false.
 * PARITY UNPINNED for this file beyond the demo near-goldens.
 * masks.bin: the upstream binary layout is unknown; this build defines its own (magic "LMMASKS1").
 */
#include "lmo.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

static uint64_t splitmix64(uint64_t *x) {
    uint64_t z = (*x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static int cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : x > y;
}

static int mask_prefix_len(int M) {
    /* lib-index-search.go:467: max(int(math.Log2(float64(M))/2), 1) */
    int p = (int)(log2((double)M) / 2);
    return p < 1 ? 1 : p;
}

/* Mask set with the structure documented at docs/content/usage/utils/masks.md:69-110: every p-base prefix is
 * present; the M-4^p extra masks reuse distinct prefixes (each at most twice) and differ from their twin in
 * base p+1 ("distinct prefixes" of length p+1); sorted ascending. Low-complexity masks are avoided. */
void lmo_gen_masks(int k, int M, int64_t seed, uint64_t *out) {
    int p = mask_prefix_len(M);
    uint64_t st = (uint64_t)seed * 0x2545F4914F6CDD1Dull + 0x1234567ull;
    int64_t np = (int64_t)1 << (p << 1);
    int lowbits = (k - p) << 1;
    uint64_t lowmask = lowbits >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << lowbits) - 1);
    int n = 0;
    if (M < np) { /* fewer masks than prefixes cannot happen with p=floor(log4 M); guard anyway */
        np = M;
    }
    for (int64_t i = 0; i < np && n < M; i++) {
        uint64_t m;
        do {
            m = ((uint64_t)i << lowbits) | (splitmix64(&st) & lowmask);
        } while (lmo_dust(m, k));
        out[n++] = m;
    }
    /* extras: choose distinct prefixes by a partial Fisher-Yates over prefix ids */
    int extra = M - n;
    if (extra > 0) {
        int64_t *perm = (int64_t *)malloc(sizeof(int64_t) * np);
        for (int64_t i = 0; i < np; i++) perm[i] = i;
        for (int j = 0; j < extra; j++) {
            int64_t r = j + (int64_t)(splitmix64(&st) % (uint64_t)(np - j));
            int64_t t = perm[j];
            perm[j] = perm[r];
            perm[r] = t;
            uint64_t twin = out[perm[j]];
            uint64_t twin_base = (twin >> (lowbits - 2)) & 3;
            uint64_t m;
            do {
                m = ((uint64_t)perm[j] << lowbits) | (splitmix64(&st) & lowmask);
            } while (((m >> (lowbits - 2)) & 3) == twin_base || lmo_dust(m, k));
            out[n++] = m;
        }
        free(perm);
    }
    qsort(out, M, sizeof(uint64_t), cmp_u64);
}

lmo_lh *lmo_lh_new(int k, const uint64_t *masks, int M) {
    lmo_lh *lh = (lmo_lh *)calloc(1, sizeof *lh);
    lh->k = k;
    lh->M = M;
    lh->p = mask_prefix_len(M);
    lh->masks = (uint64_t *)malloc(sizeof(uint64_t) * M);
    memcpy(lh->masks, masks, sizeof(uint64_t) * M);
    int64_t np = (int64_t)1 << (lh->p << 1);
    lh->pfx_first = (int *)calloc(np + 1, sizeof(int));
    int shift = (k - lh->p) << 1;
    for (int i = 0; i < M; i++) lh->pfx_first[(masks[i] >> shift) + 1]++;
    for (int64_t i = 0; i < np; i++) lh->pfx_first[i + 1] += lh->pfx_first[i];
    return lh;
}

void lmo_lh_free(lmo_lh *lh) {
    if (!lh) return;
    free(lh->masks);
    free(lh->pfx_first);
    free(lh);
}

static void put_be64(uint8_t *b, uint64_t v) {
    for (int i = 0; i < 8; i++) b[i] = (uint8_t)(v >> (56 - 8 * i));
}
static uint64_t get_be64(const uint8_t *b) {
    uint64_t v = 0;
    for (int i = 0; i < 8; i++) v = (v << 8) | b[i];
    return v;
}

/* own layout: "LMMASKS1" | k u8 | pad 3 | M u32 BE | seed i64 BE | M x u64 BE */
int lmo_lh_write(const lmo_lh *lh, const char *file, int64_t seed) {
    FILE *f = fopen(file, "wb");
    if (!f) return -1;
    uint8_t hdr[24];
    memcpy(hdr, "LMMASKS1", 8);
    hdr[8] = (uint8_t)lh->k;
    hdr[9] = hdr[10] = hdr[11] = 0;
    hdr[12] = (uint8_t)(lh->M >> 24);
    hdr[13] = (uint8_t)(lh->M >> 16);
    hdr[14] = (uint8_t)(lh->M >> 8);
    hdr[15] = (uint8_t)(lh->M);
    put_be64(hdr + 16, (uint64_t)seed);
    fwrite(hdr, 1, 24, f);
    uint8_t b[8];
    for (int i = 0; i < lh->M; i++) {
        put_be64(b, lh->masks[i]);
        fwrite(b, 1, 8, f);
    }
    fclose(f);
    return 0;
}

lmo_lh *lmo_lh_read(const char *file, int64_t *seed) {
    FILE *f = fopen(file, "rb");
    if (!f) return NULL;
    uint8_t hdr[24];
    if (fread(hdr, 1, 24, f) != 24 || memcmp(hdr, "LMMASKS1", 8)) {
        fclose(f);
        return NULL;
    }
    int k = hdr[8];
    int M = (hdr[12] << 24) | (hdr[13] << 16) | (hdr[14] << 8) | hdr[15];
    if (seed) *seed = (int64_t)get_be64(hdr + 16);
    uint64_t *m = (uint64_t *)malloc(sizeof(uint64_t) * M);
    uint8_t b[8];
    for (int i = 0; i < M; i++) {
        if (fread(b, 1, 8, f) != 8) {
            free(m);
            fclose(f);
            return NULL;
        }
        m[i] = get_be64(b);
    }
    fclose(f);
    lmo_lh *lh = lmo_lh_new(k, m, M);
    free(m);
    return lh;
}

/* lib-index-search.go:1328-1336: candidate masks = those sharing the p-prefix; first minimum wins */
int lmo_lh_mask_kmer_argmin(const lmo_lh *lh, uint64_t kmer) {
    int shift = (lh->k - lh->p) << 1;
    uint64_t pfx = kmer >> shift;
    int minj = -1;
    uint64_t minh = ~(uint64_t)0;
    for (int j = lh->pfx_first[pfx]; j < lh->pfx_first[pfx + 1]; j++) {
        uint64_t h = lh->masks[j] ^ kmer;
        if (h < minh) {
            minh = h;
            minj = j;
        }
    }
    return minj;
}

typedef struct {
    int *v;
    int n, cap;
} ivec;
static void ivec_push(ivec *a, int x) {
    if (a->n == a->cap) {
        a->cap = a->cap ? a->cap * 2 : 4;
        a->v = (int *)realloc(a->v, sizeof(int) * a->cap);
    }
    a->v[a->n++] = x;
}

static inline void consider(uint64_t *hashes, ivec *locs, int i, uint64_t mask, uint64_t kmer, int loc) {
    uint64_t h = mask ^ kmer;
    if (h < hashes[i]) {
        hashes[i] = h;
        locs[i].n = 0;
        ivec_push(&locs[i], loc);
    } else if (h == hashes[i]) {
        ivec_push(&locs[i], loc);
    }
}

int lmo_lh_mask(const lmo_lh *lh, const uint8_t *seq, int len, const int *skip, int nskip, int check_shorter,
                uint64_t *kmers, int **loc_off, int **locs_out) {
    int M = lh->M, k = lh->k;
    lmo_kiter it;
    if (lmo_kiter_init(&it, seq, len, k) != 0) return -1;
    uint64_t *hashes = (uint64_t *)malloc(sizeof(uint64_t) * M);
    ivec *locs = (ivec *)calloc(M, sizeof(ivec));
    for (int i = 0; i < M; i++) hashes[i] = ~(uint64_t)0;
    int shift = (k - lh->p) << 1;
    uint64_t kmer, rc;
    int ri = 0; /* skip-region cursor */
    while (lmo_kiter_next(&it, &kmer, &rc)) {
        int idx = it.idx;
        if (nskip > 0) {
            while (ri < nskip && idx > skip[2 * ri + 1]) ri++;
            if (ri < nskip && idx + k - 1 >= skip[2 * ri] && idx <= skip[2 * ri + 1]) continue;
        }
        uint64_t pf = kmer >> shift;
        for (int j = lh->pfx_first[pf]; j < lh->pfx_first[pf + 1]; j++)
            consider(hashes, locs, j, lh->masks[j], kmer, idx << 1);
        pf = rc >> shift;
        for (int j = lh->pfx_first[pf]; j < lh->pfx_first[pf + 1]; j++)
            consider(hashes, locs, j, lh->masks[j], rc, (idx << 1) | 1);
    }
    if (check_shorter) {
        /* masks whose p-prefix never occurred: global argmin over every k-mer */
        int nmiss = 0;
        for (int i = 0; i < M; i++)
            if (locs[i].n == 0) nmiss++;
        if (nmiss > 0) {
            int *miss = (int *)malloc(sizeof(int) * nmiss);
            int c = 0;
            for (int i = 0; i < M; i++)
                if (locs[i].n == 0) miss[c++] = i;
            lmo_kiter_init(&it, seq, len, k);
            ri = 0;
            while (lmo_kiter_next(&it, &kmer, &rc)) {
                int idx = it.idx;
                if (nskip > 0) {
                    while (ri < nskip && idx > skip[2 * ri + 1]) ri++;
                    if (ri < nskip && idx + k - 1 >= skip[2 * ri] && idx <= skip[2 * ri + 1]) continue;
                }
                for (int c2 = 0; c2 < nmiss; c2++) {
                    int i = miss[c2];
                    consider(hashes, locs, i, lh->masks[i], kmer, idx << 1);
                    consider(hashes, locs, i, lh->masks[i], rc, (idx << 1) | 1);
                }
            }
            free(miss);
        }
    }
    int *off = (int *)malloc(sizeof(int) * (M + 1));
    off[0] = 0;
    for (int i = 0; i < M; i++) {
        kmers[i] = locs[i].n ? (hashes[i] ^ lh->masks[i]) : 0;
        off[i + 1] = off[i] + locs[i].n;
    }
    int *flat = (int *)malloc(sizeof(int) * (off[M] > 0 ? off[M] : 1));
    for (int i = 0; i < M; i++) {
        memcpy(flat + off[i], locs[i].v, sizeof(int) * locs[i].n);
        free(locs[i].v);
    }
    free(locs);
    free(hashes);
    *loc_off = off;
    *locs_out = flat;
    return 0;
}

/* Window masking for seed-desert filling (lib-index-build.go:1191-1240): equivalent to
 * MaskKnownDistinctPrefixes(window, nil, false) followed by the loc2maskidx / loc2maskidxRC fill, where the mask
 * recorded for a location is the LAST (largest-index) mask that captured it.  `hashes` is an M-sized scratch array
 * that must be all-ones on entry and is restored on exit; `touched` is an M-sized scratch list. */
void lmo_lh_window_l2m(const lmo_lh *lh, const uint8_t *seq, int len, uint64_t *hashes, int *touched, int *l2m,
                       int *l2mrc) {
    int k = lh->k;
    for (int i = 0; i < len; i++) l2m[i] = l2mrc[i] = -1;
    lmo_kiter it;
    if (lmo_kiter_init(&it, seq, len, k) != 0) return;
    int shift = (k - lh->p) << 1;
    int nt = 0;
    uint64_t kmer, rc;
    while (lmo_kiter_next(&it, &kmer, &rc)) {
        for (int s = 0; s < 2; s++) {
            uint64_t x = s ? rc : kmer;
            uint64_t pf = x >> shift;
            for (int j = lh->pfx_first[pf]; j < lh->pfx_first[pf + 1]; j++) {
                uint64_t h = lh->masks[j] ^ x;
                if (h < hashes[j]) {
                    if (hashes[j] == ~(uint64_t)0) touched[nt++] = j;
                    hashes[j] = h;
                }
            }
        }
    }
    lmo_kiter_init(&it, seq, len, k);
    while (lmo_kiter_next(&it, &kmer, &rc)) {
        for (int s = 0; s < 2; s++) {
            uint64_t x = s ? rc : kmer;
            uint64_t pf = x >> shift;
            for (int j = lh->pfx_first[pf]; j < lh->pfx_first[pf + 1]; j++) {
                if ((lh->masks[j] ^ x) == hashes[j]) {
                    if (s)
                        l2mrc[it.idx] = j;
                    else
                        l2m[it.idx] = j;
                }
            }
        }
    }
    for (int i = 0; i < nt; i++) hashes[touched[i]] = ~(uint64_t)0;
}
