// tests/host_algos.cpp — compiles the product's per-work-item device algorithms (lexicmap_amd/csrc/lm_algos.h) for
// the HOST so that the CPU test-suite can check the device logic against the oracle without a GPU.
// TEST INFRASTRUCTURE ONLY: the product library never runs these on the host.
#include "../lexicmap_amd/csrc/lm_algos.h"
#include <stdlib.h>
#include <string.h>
#include <vector>

extern "C" {

uint64_t ha_revcomp(uint64_t x, int k) { return lm_revcomp(x, k); }
uint64_t ha_reverse(uint64_t x, int k) { return lm_reverse(x, k); }
int ha_dust(uint64_t x, int k) { return lm_dust(x, k); }
int ha_low_complexity(uint64_t x, int k) { return lm_low_complexity(x, k); }
int ha_base2bit(int c) { return lm_base2bit((uint8_t)c); }

// packed seed image helpers (lm_seedpack.hip / k_lookup_count): element read, word rebuild, partition range query
uint64_t ha_bits_get(const uint64_t *a, int64_t i, int w) { return lm_bits_get(a, i, w); }
// writes elements [first, first+n) of width w into the stream the way k_sp_sort_parts does (full words stored, shared
// words merged under their mask)
void ha_bits_store_range(uint64_t *a, int64_t first, int64_t n, int w, const uint64_t *elems) {
    const int64_t w0 = (first * w) >> 6, w1 = ((first + n) * w - 1) >> 6;
    for (int64_t x = w0; x <= w1; x++) {
        uint64_t m;
        const uint64_t v = lm_bits_build_word(x, first, n, w, [&](int64_t i) { return elems[i]; }, &m);
        a[x] = (a[x] & ~m) | v;
    }
}
int32_t ha_partition_range(const uint64_t *keys, int key_bits, int64_t b, int64_t e, uint64_t lrem, uint64_t rrem,
                           int64_t *first) {
    return lm_partition_range(keys, key_bits, b, e, lrem, rrem, first);
}
uint64_t ha_pack_seed_val(uint64_t g, uint64_t v64, int pos_bits) { return lm_pack_seed_val(g, v64, pos_bits); }
uint64_t ha_unpack_seed_val(uint64_t pv, uint64_t bg, int pos_bits, int dir) { return lm_unpack_seed_val(pv, bg, pos_bits, dir); }

// pseudo-alignment prefix filter: bitmap of a sorted key array the way k_build_cmp_bits fills it, and the candidate test
void ha_pa_filter_build(const uint64_t *keys, int n, int K, int log, uint32_t *bits) { // lm_pa_bits_words(log) words
    for (int i = 0; i < n; i++) lm_pa_filter_set(keys[i], K, log, [&](uint64_t w, uint32_t m) { bits[w] |= m; });
}
uint64_t ha_pa_bits_words(int log) { return lm_pa_bits_words(log); }
int ha_pa_candidate2(const uint32_t *bits, int log, uint64_t key, int p, int K) {
    return lm_pa_candidate2(bits + lm_pa_bloom_word0(log), lm_pa_bloom_log(log), bits + lm_pa_map9_word0(log), bits, log,
                            (uint32_t)(key >> ((K - p) << 1)), p);
}
int ha_pa_candidate(const uint32_t *bits, int log, uint64_t key, int p, int K) { return lm_pa_candidate(bits, log, key, p, K); }

uint64_t ha_xor_argmin(const uint64_t *a, int n, uint64_t m, int *lo, int *hi) { return lm_xor_argmin(a, n, m, lo, hi); }

uint64_t ha_pack_anchor(int qb, int len, int tb, int qrc, int trc) { return lm_pack_anchor(qb, len, tb, qrc, trc); }
void ha_unpack_anchor(uint64_t v, LmSub *out) { *out = lm_unpack_anchor(v); }

int ha_clear_sorted(LmSub *subs, int n, int k) {
    std::vector<uint8_t> marks(n + 1);
    return lm_clear_sorted(subs, n, k, marks.data());
}

float ha_chain1(const LmSub *subs, int n, float max_gap, float min_score, float max_distance, int top_chains,
                const float *gap_lut, int gap_lut_n, int32_t *chain_off, int32_t *chain_idx, int *nchains) {
    std::vector<uint64_t> msi(n + 1), s2i(n + 1);
    std::vector<int8_t> dirs(n + 1);
    std::vector<uint8_t> vis(n + 1);
    LmChainOpt o;
    o.max_gap = max_gap;
    o.min_score = min_score;
    o.max_distance = max_distance;
    o.top_chains = top_chains;
    o.gap_lut = gap_lut;
    o.gap_lut_n = gap_lut_n;
    return lm_run_chain1(subs, n, o, msi.data(), s2i.data(), dirs.data(), vis.data(), chain_off, chain_idx, nchains);
}

int ha_trim(const LmSub *subs, int n, float min_dist, int *start) { return lm_trim(subs, n, min_dist, start); }

int ha_chain2(const LmSub *subs, int n, int max_gap, int min_score, int min_align_len, int band_count, int band_base,
              double hpt, LmChain2 *out) {
    std::vector<uint64_t> msi(n + 1);
    std::vector<int32_t> stack(2 * (n + 2));
    LmChain2Opt o;
    o.max_gap = max_gap;
    o.min_score = min_score;
    o.min_align_len = min_align_len;
    o.band_count = band_count;
    o.band_base = band_base;
    o.heuristic_pident = hpt;
    return lm_run_chain2(subs, n, o, msi.data(), stack.data(), out);
}

void ha_extend_match(const uint8_t *seq1, int len1, const uint8_t *seq2, int len2, int start1, int end1, int start2,
                     int end2, int ext_len, int tbegin, int max_ext_len, int rc, int *o) {
    int cap = 200 * 200;
    std::vector<LmSub> subs(cap);
    std::vector<int64_t> msi(cap);
    lm_extend_match(seq1, len1, seq2, len2, start1, end1, start2, end2, ext_len, tbegin, max_ext_len, rc != 0,
                    subs.data(), msi.data(), cap, &o[0], &o[1], &o[2], &o[3], &o[4], &o[5], &o[6], &o[7]);
}

// one flank through both forms of the 2-mer chainer: list (lm_extend_right) and grid (lm_extend_flank_grid, the form the
// HIP kernel runs, here with a stride of 3 to exercise the interleaved scratch addressing)
void ha_extend_flank_both(const uint8_t *s1, int n1, const uint8_t *s2, int n2, int rev, int *o) {
    int cap = 260 * 260;
    std::vector<LmSub> subs(cap);
    std::vector<int64_t> msi(cap);
    lm_extend_right(s1, n1, s2, n2, rev != 0, subs.data(), msi.data(), cap, &o[0], &o[1]);
    const int stride = 3;
    std::vector<uint16_t> subs16((size_t)cap * stride, 0xffff);
    std::vector<int32_t> msi32((size_t)cap * stride, -1);
    std::vector<LmM128> rows((size_t)LM_EXT_ROWS * stride);
    std::vector<uint32_t> rstart((size_t)LM_EXT_ROWS * stride);
    lm_extend_flank_grid(s1, n1, s2, n2, rev != 0, subs16.data() + 1, msi32.data() + 1, cap, rows.data() + 1, rstart.data() + 1,
                         stride, &o[2], &o[3]);
}

int ha_tree_search_range(const uint64_t *keys, int n, uint64_t key, int p, int K, int *lo, int *hi) {
    return lm_tree_search_range(keys, n, key, p, K, lo, hi) ? 1 : 0;
}

// the lower-bound-only form of k_pa_search over a bucket table built the way k_build_cmp_tab builds it; the enumeration
// "while keys[j] <= right" is done here so that the result compares directly with ha_tree_search_range
void ha_build_tab(const uint64_t *keys, int n, int K, int tab_bits, uint32_t *tab) { // (1 << tab_bits) + 1 entries
    for (uint32_t b = 0; b <= (1u << tab_bits); b++)
        tab[b] = b == (1u << tab_bits) ? (uint32_t)n
                                        : (uint32_t)lm_lower_bound_u64(keys, 0, n, (uint64_t)b << ((K << 1) - tab_bits));
}
int ha_tree_search_first_tab(const uint64_t *keys, int n, uint64_t key, int p, int K, const uint32_t *tab, int tab_bits,
                             int *lo, int *hi) {
    uint64_t right = 0;
    int r = lm_tree_search_first_tab(keys, n, key, p, K, tab, tab_bits, lo, hi, &right);
    if (r == 1) {
        int j = *lo;
        while (j < *hi && keys[j] <= right) j++;
        *hi = j;
    }
    return r != 0 ? 1 : 0;
}

// returns status; ops copied out
int ha_wfa(const uint8_t *q, int plen, const uint8_t *t, int tlen, int max_score, int64_t arena_cap, uint64_t *ops,
           int ops_cap, LmWfaOut *out) {
    std::vector<int32_t> hdr((size_t)9 * max_score + 9);
    std::vector<int32_t> arena((size_t)arena_cap);
    lm_wfa_align(q, plen, t, tlen, hdr.data(), max_score, arena.data(), arena_cap, ops, ops_cap, out);
    return out->status;
}
}
