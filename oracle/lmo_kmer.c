/* CPU ORACLE (test infrastructure only) — k-mer bit utilities.
 * Follows util/kmers.go, genome/genome.go:1427-1444 and the lexichash/iterator call sites
 * (lib-seq_compare.go:120,138; lib-index-build.go:1191-1205). */
#include "lmo.h"
#include <string.h>

/* genome/genome.go:1427-1444 — A=0 C=1 G=2 T/U=3; B,S,Y->1; K->2; everything else -> 0 */
const uint8_t lmo_base2bit[256] = {
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
    0, 0, 1, 1, 0, 0, 0, 2, 0, 0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 1, 3, 3, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0,
    0, 0, 1, 1, 0, 0, 0, 2, 0, 0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 1, 3, 3, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0,
};

uint64_t lmo_kmer_encode(const uint8_t *s, int k) {
    uint64_t c = 0;
    for (int i = 0; i < k; i++) c = (c << 2) | lmo_base2bit[s[i]];
    return c;
}

void lmo_kmer_decode(uint64_t code, int k, char *out) {
    static const char b[4] = {'A', 'C', 'G', 'T'};
    for (int i = 0; i < k; i++) out[i] = b[(code >> ((k - 1 - i) << 1)) & 3];
    out[k] = 0;
}

uint64_t lmo_kmer_revcomp(uint64_t code, int k) {
    uint64_t r = 0;
    for (int i = 0; i < k; i++) {
        r = (r << 2) | (3 - (code & 3));
        code >>= 2;
    }
    return r;
}

/* kmers.MustReverse (kmers v0.1.0): base-wise reversal, no complement (doc example usage/utils/kmers.md:114-124) */
uint64_t lmo_kmer_reverse(uint64_t code, int k) {
    uint64_t r = 0;
    for (int i = 0; i < k; i++) {
        r = (r << 2) | (code & 3);
        code >>= 2;
    }
    return r;
}

/* util/kmers.go:434-441 */
uint64_t lmo_ns(uint64_t b, int k) {
    uint64_t code = b;
    for (int i = 1; i < k; i++) code = (code << 2) + b;
    return code;
}

/* util/kmers.go:162-328: 3-mer windows at shifts i=0..k-2 (the last window reaches 2 bits above the k-mer,
 * which are zero); score = sum c(c-1)/2 in uint16; low complexity iff score > 50 */
int lmo_dust(uint64_t code, int k) {
    uint8_t counts[64];
    memset(counts, 0, sizeof counts);
    int end = k - 2;
    for (int i = 0; i <= end; i++) counts[(code >> (i << 1)) & 63]++;
    uint16_t score = 0;
    for (int i = 0; i < 64; i++) {
        uint16_t c = counts[i];
        score += (uint16_t)((uint16_t)(c - 1) * c) >> 1;
    }
    return score > 50;
}

/* lib-index-search.go:1223-1238 (also lib-index-build.go:1033-1046, lib-seq_compare.go:143) */
int lmo_low_complexity(uint64_t kmer, int k) {
    uint64_t ccc = lmo_ns(1, k), ggg = lmo_ns(2, k);
    uint64_t ttt = ((uint64_t)1 << (k << 1)) - 1;
    return kmer == ccc || kmer == ggg || kmer == ttt || lmo_dust(kmer, k);
}

int lmo_kiter_init(lmo_kiter *it, const uint8_t *s, int len, int k) {
    if (len < k || k < 1 || k > 32) return -1;
    it->s = s;
    it->len = len;
    it->k = k;
    it->idx = -1;
    it->fwd = it->rc = 0;
    it->mask = k == 32 ? ~(uint64_t)0 : (((uint64_t)1 << (k << 1)) - 1);
    it->started = 0;
    return 0;
}

/* returns 1 and the forward / reverse-complement k-mer of the next window; Index() == it->idx */
int lmo_kiter_next(lmo_kiter *it, uint64_t *kmer, uint64_t *kmer_rc) {
    int k = it->k;
    if (!it->started) {
        it->started = 1;
        it->fwd = lmo_kmer_encode(it->s, k);
        it->rc = lmo_kmer_revcomp(it->fwd, k);
        it->idx = 0;
    } else {
        if (it->idx + k >= it->len) return 0;
        uint64_t b = lmo_base2bit[it->s[it->idx + k]];
        it->fwd = ((it->fwd << 2) | b) & it->mask;
        it->rc = (it->rc >> 2) | ((3 - b) << ((k - 1) << 1));
        it->idx++;
    }
    *kmer = it->fwd;
    if (kmer_rc) *kmer_rc = it->rc;
    return 1;
}
