/* CPU ORACLE (test infrastructure only) — radix tree over 2-bit packed k-mers.
 * Follows tree/tree.go: node/leaf layout :30-60, Insert/InsertBatch :142-380 (the compressed trie over a key set is
 * canonical, so it is built here from the sorted distinct keys), Search :441-527 (restated literally, including the
 * uint8 shift arithmetic of the partial-prefix check at :496-500), recursiveWalk :543-555.
 * Pinned by tree/tree_test.go:72-248 (InsertBatch == repeated Insert; Search vs brute-force LCP) restated in tests/. */
#include "lmo.h"
#include <stdlib.h>
#include <string.h>

typedef struct tnode {
    uint64_t prefix;
    uint8_t k;
    struct tnode *children[4];
    int leaf; /* index into leaves or -1 */
} tnode;

typedef struct {
    uint64_t key;
    uint32_t *vals;
    int n, cap;
} tleaf;

struct lmo_tree {
    int k;
    tnode *root;
    tnode **chunks;
    int nchunks, pos;
    tleaf *leaves;
    int nleaves, capleaves;
    /* pending entries for lazily (re)building */
    lmo_tree_entry *ent;
    int nent, capent;
    int dirty;
};

#define CHUNK 1024

static tnode *node_alloc(lmo_tree *t) {
    if (t->nchunks == 0 || t->pos >= CHUNK) {
        t->chunks = (tnode **)realloc(t->chunks, sizeof(tnode *) * (t->nchunks + 1));
        t->chunks[t->nchunks++] = (tnode *)malloc(sizeof(tnode) * CHUNK);
        t->pos = 0;
    }
    tnode *n = &t->chunks[t->nchunks - 1][t->pos++];
    memset(n, 0, sizeof *n);
    n->leaf = -1;
    return n;
}

lmo_tree *lmo_tree_new(int k) {
    lmo_tree *t = (lmo_tree *)calloc(1, sizeof *t);
    t->k = k;
    return t;
}

static void tree_clear_nodes(lmo_tree *t) {
    for (int i = 0; i < t->nchunks; i++) free(t->chunks[i]);
    free(t->chunks);
    t->chunks = NULL;
    t->nchunks = 0;
    t->pos = 0;
    for (int i = 0; i < t->nleaves; i++) free(t->leaves[i].vals);
    free(t->leaves);
    t->leaves = NULL;
    t->nleaves = t->capleaves = 0;
    t->root = NULL;
}

void lmo_tree_free(lmo_tree *t) {
    if (!t) return;
    tree_clear_nodes(t);
    free(t->ent);
    free(t);
}

void lmo_tree_insert(lmo_tree *t, uint64_t key, uint32_t v) {
    if (t->nent == t->capent) {
        t->capent = t->capent ? t->capent * 2 : 64;
        t->ent = (lmo_tree_entry *)realloc(t->ent, sizeof(lmo_tree_entry) * t->capent);
    }
    t->ent[t->nent].key = key;
    t->ent[t->nent].val = v;
    t->nent++;
    t->dirty = 1;
}

/* stable merge sort by key */
static void msort(lmo_tree_entry *a, lmo_tree_entry *tmp, int n) {
    if (n < 2) return;
    int h = n / 2;
    msort(a, tmp, h);
    msort(a + h, tmp, n - h);
    int i = 0, j = h, o = 0;
    while (i < h && j < n) tmp[o++] = (a[j].key < a[i].key) ? a[j++] : a[i++];
    while (i < h) tmp[o++] = a[i++];
    while (j < n) tmp[o++] = a[j++];
    memcpy(a, tmp, sizeof(lmo_tree_entry) * n);
}

static inline int lcp(uint64_t a, uint64_t b, int k) {
    uint64_t x = a ^ b;
    int lz = x ? __builtin_clzll(x) : 64;
    return (lz >> 1) + k - 32;
}
static inline int base_at(uint64_t code, int k, int i) { return (int)((code >> ((k - i - 1) << 1)) & 3); }

/* leaves[lo..hi) are distinct sorted keys sharing their first `depth` bases */
static void build_children(lmo_tree *t, tnode *parent, int lo, int hi, int depth) {
    int K = t->k;
    int a = lo;
    while (a < hi) {
        int c = base_at(t->leaves[a].key, K, depth);
        int b = a + 1;
        while (b < hi && base_at(t->leaves[b].key, K, depth) == c) b++;
        int edge_end = (b - a == 1) ? K : lcp(t->leaves[a].key, t->leaves[b - 1].key, K);
        tnode *n = node_alloc(t);
        n->k = (uint8_t)(edge_end - depth);
        /* bases [depth, edge_end) of the key */
        uint64_t suffix = depth == 0 ? t->leaves[a].key
                                     : (t->leaves[a].key & ((((uint64_t)1) << ((K - depth) << 1)) - 1));
        n->prefix = suffix >> ((K - edge_end) << 1);
        parent->children[c] = n;
        if (edge_end == K)
            n->leaf = a;
        else
            build_children(t, n, a, b, edge_end);
        a = b;
    }
}

static void rebuild(lmo_tree *t) {
    tree_clear_nodes(t);
    t->root = node_alloc(t);
    if (t->nent > 0) {
        lmo_tree_entry *tmp = (lmo_tree_entry *)malloc(sizeof(lmo_tree_entry) * t->nent);
        msort(t->ent, tmp, t->nent);
        free(tmp);
        for (int i = 0; i < t->nent; i++) {
            if (t->nleaves == 0 || t->leaves[t->nleaves - 1].key != t->ent[i].key) {
                if (t->nleaves == t->capleaves) {
                    t->capleaves = t->capleaves ? t->capleaves * 2 : 64;
                    t->leaves = (tleaf *)realloc(t->leaves, sizeof(tleaf) * t->capleaves);
                }
                tleaf *l = &t->leaves[t->nleaves++];
                l->key = t->ent[i].key;
                l->vals = NULL;
                l->n = l->cap = 0;
            }
            tleaf *l = &t->leaves[t->nleaves - 1];
            if (l->n == l->cap) {
                l->cap = l->cap ? l->cap * 2 : 2;
                l->vals = (uint32_t *)realloc(l->vals, sizeof(uint32_t) * l->cap);
            }
            l->vals[l->n++] = t->ent[i].val;
        }
        build_children(t, t->root, 0, t->nleaves, 0);
    }
    t->dirty = 0;
}

void lmo_tree_insert_batch(lmo_tree *t, lmo_tree_entry *e, int n) {
    for (int i = 0; i < n; i++) lmo_tree_insert(t, e[i].key, e[i].val);
    rebuild(t);
}

typedef struct {
    const lmo_tree *t;
    lmo_tree_sr **out;
    int *cap;
    int n;
    uint64_t key0;
    int k0;
} walkctx;

static void walk(const tnode *n, walkctx *w) {
    if (n->leaf >= 0) {
        if (w->n == *w->cap) {
            *w->cap = *w->cap ? *w->cap * 2 : 16;
            *w->out = (lmo_tree_sr *)realloc(*w->out, sizeof(lmo_tree_sr) * *w->cap);
        }
        const tleaf *l = &w->t->leaves[n->leaf];
        lmo_tree_sr *r = &(*w->out)[w->n++];
        r->kmer = l->key;
        r->len_prefix = (uint8_t)lcp(w->key0, l->key, w->k0);
        r->vals = l->vals;
        r->nvals = l->n;
    }
    for (int c = 0; c < 4; c++)
        if (n->children[c]) walk(n->children[c], w);
}

/* tree.go:441-527 */
int lmo_tree_search(const lmo_tree *tc, uint64_t key, int p_in, lmo_tree_sr **out, int *cap) {
    lmo_tree *t = (lmo_tree *)tc;
    if (t->dirty || !t->root) rebuild(t);
    uint8_t p = (uint8_t)p_in;
    if (p < 1) p = 1;
    uint8_t k = (uint8_t)t->k;
    if (p > k) p = k;
    uint64_t key0 = key;
    uint8_t k0 = k;
    const tnode *target = NULL;
    const tnode *n = t->root;
    uint64_t search = key;
    uint8_t len_prefix = 0, atleast;
    for (;;) {
        if (k == 0) break;
        n = n->children[base_at(search, k, 0)];
        if (n == NULL) break;
        if ((search >> ((k - n->k) << 1)) == n->prefix) { /* MustKmerHasPrefix */
            len_prefix = (uint8_t)(len_prefix + n->k);
            if (len_prefix >= p) {
                target = n;
                break;
            }
            search = search & ((((uint64_t)1) << ((k - n->k) << 1)) - 1); /* KmerSuffix(search,k,n.k) */
            k = (uint8_t)(k - n->k);
        } else {
            atleast = (uint8_t)(p - len_prefix);
            /* Go: search>>((k-atleast)<<1) == n.prefix>>((n.k-atleast)<<1); operands are uint8 and wrap;
             * a shift count >= 64 yields 0 */
            uint8_t s1 = (uint8_t)((uint8_t)(k - atleast) << 1);
            uint8_t s2 = (uint8_t)((uint8_t)(n->k - atleast) << 1);
            uint64_t lhs = s1 >= 64 ? 0 : (search >> s1);
            uint64_t rhs = s2 >= 64 ? 0 : (n->prefix >> s2);
            if (lhs == rhs) target = n;
            break;
        }
    }
    if (target == NULL) return 0;
    walkctx w = {t, out, cap, 0, key0, k0};
    walk(target, &w);
    return w.n;
}
