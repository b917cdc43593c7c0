/* CPU ORACLE (test infrastructure only) — pseudo-alignment and HSP end extension.
 * Follows lib-seq_compare.go:115-159 (Index), :270-308 (coverageLen), :335-522 (Compare),
 * lib-index-search-util.go:34-201 (extendMatch/_extendRight). */
#include "lmo.h"
#include <stdlib.h>
#include <string.h>

lmo_cmp *lmo_cmp_new(const lmo_cmp_opt *opt) {
    lmo_cmp *c = (lmo_cmp *)calloc(1, sizeof *c);
    c->opt = *opt;
    return c;
}

void lmo_cmp_free(lmo_cmp *c) {
    if (!c) return;
    lmo_tree_free(c->tree);
    free(c);
}

/* lib-seq_compare.go:115-159 */
int lmo_cmp_index(lmo_cmp *c, const uint8_t *s, int len) {
    int k = c->opt.k;
    lmo_kiter it;
    if (lmo_kiter_init(&it, s, len, k) != 0) return -1;
    lmo_tree_free(c->tree);
    c->tree = lmo_tree_new(k);
    uint64_t ccc = lmo_ns(1, k), ggg = lmo_ns(2, k), ttt = lmo_ns(3, k);
    int cap = 2 * (len - k + 1), n = 0;
    lmo_tree_entry *e = (lmo_tree_entry *)malloc(sizeof(lmo_tree_entry) * (cap > 0 ? cap : 1));
    uint64_t kmer, rc;
    while (lmo_kiter_next(&it, &kmer, &rc)) {
        if (kmer == 0 || kmer == ccc || kmer == ggg || kmer == ttt || lmo_dust(kmer, k)) continue;
        e[n].key = kmer;
        e[n].val = (uint32_t)(it.idx << 1);
        n++;
        e[n].key = rc;
        e[n].val = (uint32_t)((it.idx << 1) | 1);
        n++;
    }
    lmo_tree_insert_batch(c->tree, e, n);
    free(e);
    return 0;
}

static int cmp_region(const void *a, const void *b) {
    const int *x = (const int *)a, *y = (const int *)b;
    return x[0] < y[0] ? -1 : x[0] > y[0];
}

/* lib-seq_compare.go:270-308 */
int lmo_coverage_len(int (*regions)[2], int n) {
    if (n == 0) return 0;
    if (n == 1) return regions[0][1] - regions[0][0] + 1;
    qsort(regions, n, sizeof(int[2]), cmp_region);
    int r = 0;
    int start = regions[0][0], end = regions[0][1];
    for (int i = 1; i < n; i++) {
        if (regions[i][0] > end) {
            r += end - start + 1;
            start = regions[i][0];
            end = regions[i][1];
            continue;
        }
        if (regions[i][1] <= end) continue;
        end = regions[i][1];
    }
    r += end - start + 1;
    return r;
}

typedef struct {
    lmo_sub *v;
    int n, cap;
} subvec;
static lmo_sub *subpush(subvec *s) {
    if (s->n == s->cap) {
        s->cap = s->cap ? s->cap * 2 : 256;
        s->v = (lmo_sub *)realloc(s->v, sizeof(lmo_sub) * s->cap);
    }
    return &s->v[s->n++];
}

static int cmp_chain2_qbegin(const void *a, const void *b) {
    const lmo_chain2 *x = (const lmo_chain2 *)a, *y = (const lmo_chain2 *)b;
    return x->qbegin < y->qbegin ? -1 : x->qbegin > y->qbegin;
}

/* lib-seq_compare.go:335-522 */
int lmo_cmp_compare(lmo_cmp *c, uint32_t begin, uint32_t end, const uint8_t *s, int slen, int query_len,
                    lmo_chain2 **chains_out, lmo_sub **subs_out, int *nsubs_out) {
    (void)query_len;
    *chains_out = NULL;
    if (subs_out) {
        *subs_out = NULL;
        *nsubs_out = 0;
    }
    int k = c->opt.k;
    int m = c->opt.min_prefix;
    if (slen >= 1000000)
        m += 8;
    else if (slen >= 250000)
        m += 6;
    else if (slen >= 50000)
        m += 4;
    else if (slen >= 10000)
        m += 2;
    lmo_kiter it;
    if (lmo_kiter_init(&it, s, slen, k) != 0) return 0;
    uint64_t ccc = lmo_ns(1, k), ggg = lmo_ns(2, k), ttt = lmo_ns(3, k);
    subvec subs = {0};
    lmo_tree_sr *srs = NULL;
    int srcap = 0;
    uint64_t kmer, rc;
    while (lmo_kiter_next(&it, &kmer, &rc)) {
        if (kmer == 0 || kmer == ccc || kmer == ggg || kmer == ttt) continue;
        int ns = lmo_tree_search(c->tree, kmer, m, &srs, &srcap);
        for (int i = 0; i < ns; i++) {
            for (int j = 0; j < srs[i].nvals; j++) {
                uint32_t v = srs[i].vals[j];
                uint32_t p = v >> 1;
                if ((v & 1) == 1 || p < begin || p + (uint32_t)srs[i].len_prefix > end) continue;
                lmo_sub *sb = subpush(&subs);
                sb->qbegin = (int32_t)p;
                sb->tbegin = (int32_t)it.idx;
                sb->len = srs[i].len_prefix;
                sb->qrc = 0;
                sb->trc = 0;
                sb->_pad = 0;
            }
        }
        ns = lmo_tree_search(c->tree, rc, m, &srs, &srcap);
        for (int i = 0; i < ns; i++) {
            for (int j = 0; j < srs[i].nvals; j++) {
                uint32_t v = srs[i].vals[j];
                uint32_t p = (v >> 1) + (uint32_t)k - (uint32_t)srs[i].len_prefix;
                if ((v & 1) == 0 || p + (uint32_t)srs[i].len_prefix < begin || p > end) continue;
                lmo_sub *sb = subpush(&subs);
                sb->qbegin = (int32_t)p;
                sb->tbegin = (int32_t)(it.idx + k - (int)srs[i].len_prefix);
                sb->len = srs[i].len_prefix;
                sb->qrc = 1;
                sb->trc = 1;
                sb->_pad = 0;
            }
        }
    }
    free(srs);
    if (subs.n < 1) {
        free(subs.v);
        return 0;
    }
    if (subs.n > 1) subs.n = lmo_clear_subs(subs.v, subs.n, k);
    subs.n = lmo_trim_subs(subs.v, subs.n, k, 100, NULL);
    if (subs.n == 0) {
        free(subs.v);
        return 0;
    }
    lmo_chain2 *chains = NULL;
    int aq = 0;
    int nc = lmo_chainer2(subs.v, subs.n, &c->opt.c2, &chains, &aq);
    if (subs_out) {
        *subs_out = subs.v;
        *nsubs_out = subs.n;
    } else {
        free(subs.v);
    }
    if (nc == 0) return 0;
    /* "very important": sort by QBegin.  slices.SortFunc is insertion sort (stable) up to 12 elements; the oracle
     * uses a stable order for every size (ties keep chainARegion emission order). */
    if (nc > 1) {
        /* stable insertion sort */
        for (int i = 1; i < nc; i++) {
            lmo_chain2 x = chains[i];
            int j = i - 1;
            while (j >= 0 && chains[j].qbegin > x.qbegin) {
                chains[j + 1] = chains[j];
                j--;
            }
            chains[j + 1] = x;
        }
    }
    (void)cmp_chain2_qbegin;
    *chains_out = chains;
    return nc;
}

/* ------------------------------------------------------------------------------------------------
 * extendMatch, lib-index-search-util.go:34-201 */
static void extend_right(const uint8_t *s1, int n1, const uint8_t *s2, int n2, int *o1, int *o2) {
    *o1 = 0;
    *o2 = 0;
    const int _k = 2, m = 2;
    lmo_kiter it;
    if (lmo_kiter_init(&it, s1, n1, _k) != 0) return;
    lmo_tree *t = lmo_tree_new(_k);
    uint64_t kmer;
    while (lmo_kiter_next(&it, &kmer, NULL)) lmo_tree_insert(t, kmer, (uint32_t)it.idx);
    if (lmo_kiter_init(&it, s2, n2, _k) != 0) {
        lmo_tree_free(t);
        return;
    }
    subvec subs = {0};
    lmo_tree_sr *srs = NULL;
    int srcap = 0;
    while (lmo_kiter_next(&it, &kmer, NULL)) {
        int ns = lmo_tree_search(t, kmer, m, &srs, &srcap);
        for (int i = 0; i < ns; i++)
            for (int j = 0; j < srs[i].nvals; j++) {
                lmo_sub *sb = subpush(&subs);
                sb->qbegin = (int32_t)srs[i].vals[j];
                sb->tbegin = (int32_t)it.idx;
                sb->len = srs[i].len_prefix;
                sb->qrc = sb->trc = sb->_pad = 0;
            }
    }
    free(srs);
    lmo_tree_free(t);
    if (subs.n == 0) {
        free(subs.v);
        return;
    }
    if (subs.n > 1) lmo_sort_subs(subs.v, subs.n);
    int qe, te;
    if (lmo_chainer3(subs.v, subs.n, &qe, &te)) {
        *o1 = qe + 1;
        *o2 = te + 1;
    }
    free(subs.v);
}

static void reverse_copy(const uint8_t *s, int n, uint8_t *out) {
    for (int i = 0; i < n; i++) out[i] = s[n - 1 - i];
}

void lmo_extend_match(const uint8_t *seq1, int len1, const uint8_t *seq2, int len2, int start1, int end1, int start2,
                      int end2, int ext_len, int tbegin, int max_ext_len, int rc, int *o_start1, int *o_end1,
                      int *o_start2, int *o_end2, int *s1o, int *e1o, int *s2o, int *e2o) {
    const int m = 2;
    int _start1 = start1, _end1 = end1, _start2 = start2, _end2 = end2;
    int _s1 = 0, _e1 = 0, _s2 = 0, _e2 = 0, _ext;
    if (end1 + m < len1 && end2 + m < len2) {
        _ext = rc ? (ext_len < tbegin ? ext_len : tbegin) : (ext_len < max_ext_len ? ext_len : max_ext_len);
        if (_ext > 2) {
            int e1 = end1 + _ext < len1 ? end1 + _ext : len1;
            int e2 = end2 + _ext < len2 ? end2 + _ext : len2;
            extend_right(seq1 + end1, e1 - end1, seq2 + end2, e2 - end2, &_e1, &_e2);
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
            int n1 = start1 - s1, n2 = start2 - s2;
            uint8_t *r1 = (uint8_t *)malloc(n1 + 1), *r2 = (uint8_t *)malloc(n2 + 1);
            reverse_copy(seq1 + s1, n1, r1);
            reverse_copy(seq2 + s2, n2, r2);
            extend_right(r1, n1, r2, n2, &_s1, &_s2);
            if (_s1 > 0 || _s2 > 0) {
                start1 -= _s1;
                start2 -= _s2;
            }
            free(r1);
            free(r2);
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
