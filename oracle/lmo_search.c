/* CPU ORACLE (test infrastructure only) — the `lexicmap search` pipeline for one query.
 * Follows lib-index-search.go: NewIndexSearcher :237-757 (in-RAM regime, -w), Search :1191-2940,
 * search.go:296-382 (option wiring), :468-523 (row printer).
 *
 * Deterministic choices where the reference is order-dependent (goroutine arrival / unstable sorts), all documented in
 * anchors of a genome are totally ordered before ClearSubstrPairs; chains of a genome are stably sorted by
 * the first anchor's TBegin; HSP clusters of a genome are stably sorted by SimilarityScore desc (ties: processing
 * order = chain order); genomes are sorted by best SimilarityScore desc, ties by (batch<<17|genome) asc.
 */
#include "lmo.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <dirent.h>

void lmo_search_opt_default(lmo_search_opt *o) { /* search.go:631-731 */
    o->min_prefix = 15;
    o->min_single_prefix = 17;
    o->top_n = 0;
    o->top_n_chains = 0;
    o->max_gap = 50;
    o->max_distance = 1000;
    o->ext_len = 1000;
    o->ext_len2 = 50;
    o->min_qcov_genome = 0;
    o->max_evalue = 10;
    o->output_seq = 0;
    o->align_max_gap = 20;
    o->align_band = 100;
    o->align_min_match_len = 50;
    o->align_min_pident = 70;
    o->min_qcov_hsp = 0;
}

typedef struct {
    uint64_t key;
    char *id;
} idmap_ent;

struct lmo_index {
    char *dir;
    lmo_search_opt opt;
    lmo_lh *lh;
    int k, M, mask_prefix, anchor_prefix;
    int nchunks;
    lmo_kv_mem **chunks;
    int ngbatches;
    lmo_greader **grdr;
    idmap_ent *ids;
    int nids;
    int64_t total_bases;
    int contig_interval;
    lmo_cmp_opt cmpopt;
    float chain_min_score;
    /* genome chunks (genomes.chunks.bin, lib-index-search.go:504-538): key -> list number, #chunks, chunk index */
    int has_chunks, nchunk_ent;
    struct chunk_ent {
        uint64_t key;
        int list, n, idx;
    } *chunk_ents; /* sorted by key */
};

static int cmp_chunk_ent(const void *a, const void *b) {
    uint64_t x = ((const struct chunk_ent *)a)->key, y = ((const struct chunk_ent *)b)->key;
    return x < y ? -1 : x > y;
}
static const struct chunk_ent *find_chunk(const lmo_index *idx, uint64_t key) {
    int lo = 0, hi = idx->nchunk_ent;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (idx->chunk_ents[mid].key < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo < idx->nchunk_ent && idx->chunk_ents[lo].key == key ? &idx->chunk_ents[lo] : NULL;
}

int lmo_index_k(const lmo_index *idx) { return idx->k; }
int lmo_index_nmasks(const lmo_index *idx) { return idx->M; }
const uint64_t *lmo_index_masks(const lmo_index *idx) { return idx->lh->masks; }
int64_t lmo_index_total_bases(const lmo_index *idx) { return idx->total_bases; }
const lmo_lh *lmo_index_lh(const lmo_index *idx) { return idx->lh; }

static long long toml_int(const char *text, const char *key, long long dflt) {
    size_t kl = strlen(key);
    const char *p = text;
    while (p && *p) {
        const char *eol = strchr(p, '\n');
        if (strncmp(p, key, kl) == 0) {
            const char *q = p + kl;
            while (*q == ' ') q++;
            if (*q == '=') return atoll(q + 1);
        }
        p = eol ? eol + 1 : NULL;
    }
    return dflt;
}

static int cmp_str(const void *a, const void *b) { return strcmp(*(char *const *)a, *(char *const *)b); }
static int cmp_idmap(const void *a, const void *b) {
    uint64_t x = ((const idmap_ent *)a)->key, y = ((const idmap_ent *)b)->key;
    return x < y ? -1 : x > y;
}
static int cmp_chunk(const void *a, const void *b) {
    const lmo_kv_mem *x = *(lmo_kv_mem *const *)a, *y = *(lmo_kv_mem *const *)b;
    return x->chunk_index < y->chunk_index ? -1 : x->chunk_index > y->chunk_index;
}

lmo_index *lmo_index_open(const char *dir, const lmo_search_opt *opt) {
    char p[4096];
    snprintf(p, sizeof p, "%s/info.toml", dir);
    FILE *f = fopen(p, "rb");
    if (!f) return NULL;
    char text[8192];
    size_t nr = fread(text, 1, sizeof text - 1, f);
    text[nr] = 0;
    fclose(f);
    if (toml_int(text, "main-version", -1) != 3) return NULL; /* :290-292 */
    /* :1212-1215: format < 3.5 is masked with MaskKnownDistinctPrefixesWithStrandBias (un-vendored lexichash module, not
     * restated): such an index is refused rather than searched with the wrong masking */
    if (toml_int(text, "minor-version", 0) < 5) return NULL;
    lmo_index *idx = (lmo_index *)calloc(1, sizeof *idx);
    idx->dir = strdup(dir);
    idx->opt = *opt;
    idx->total_bases = toml_int(text, "input-bases", 0);
    idx->contig_interval = (int)toml_int(text, "contig-interval", 1000);
    int partitions = (int)toml_int(text, "index-partitions", 4096);
    idx->ngbatches = (int)toml_int(text, "genome-batches", 1);
    snprintf(p, sizeof p, "%s/masks.bin", dir);
    idx->lh = lmo_lh_read(p, NULL);
    if (!idx->lh) {
        lmo_index_close(idx);
        return NULL;
    }
    idx->k = idx->lh->k;
    idx->M = idx->lh->M;
    idx->mask_prefix = idx->lh->p;
    {
        int ap = (int)(log2((double)partitions) / 2);
        idx->anchor_prefix = ap < 1 ? 1 : ap;
    }
    /* seeds */
    snprintf(p, sizeof p, "%s/seeds", dir);
    DIR *d = opendir(p);
    if (!d) {
        lmo_index_close(idx);
        return NULL;
    }
    char **names = NULL;
    int nn = 0;
    struct dirent *de;
    while ((de = readdir(d))) {
        size_t l = strlen(de->d_name);
        if (l > 4 && strcmp(de->d_name + l - 4, ".bin") == 0) {
            names = (char **)realloc(names, sizeof(char *) * (nn + 1));
            names[nn++] = strdup(de->d_name);
        }
    }
    closedir(d);
    qsort(names, nn, sizeof(char *), cmp_str);
    idx->chunks = (lmo_kv_mem **)calloc(nn ? nn : 1, sizeof(lmo_kv_mem *));
    for (int i = 0; i < nn; i++) {
        snprintf(p, sizeof p, "%s/seeds/%s", dir, names[i]);
        lmo_kv_mem *m = lmo_kv_load(p);
        free(names[i]);
        if (!m) continue;
        if (m->anchor_prefix != idx->anchor_prefix) idx->anchor_prefix = m->anchor_prefix; /* :611-612 */
        idx->chunks[idx->nchunks++] = m;
    }
    free(names);
    qsort(idx->chunks, idx->nchunks, sizeof(lmo_kv_mem *), cmp_chunk);
    if (idx->nchunks == 0 || opt->min_prefix > idx->k || opt->min_prefix < idx->mask_prefix + idx->anchor_prefix) {
        lmo_index_close(idx); /* :483-485 */
        return NULL;
    }
    /* genomes */
    idx->grdr = (lmo_greader **)calloc(idx->ngbatches, sizeof(lmo_greader *));
    for (int i = 0; i < idx->ngbatches; i++) {
        snprintf(p, sizeof p, "%s/genomes/batch_%04d/genomes.bin", dir, i);
        idx->grdr[i] = lmo_greader_open(p);
    }
    /* genome id map (lib-index-build.go:1969-2016) */
    snprintf(p, sizeof p, "%s/genomes.map.bin", dir);
    f = fopen(p, "rb");
    if (f) {
        uint8_t b[8];
        int cap = 0;
        while (fread(b, 1, 2, f) == 2) {
            int l = (b[0] << 8) | b[1];
            char *id = (char *)malloc(l + 1);
            if ((int)fread(id, 1, l, f) != l) break;
            id[l] = 0;
            if (fread(b, 1, 8, f) != 8) break;
            uint64_t key = 0;
            for (int i = 0; i < 8; i++) key = (key << 8) | b[i];
            if (idx->nids == cap) {
                cap = cap ? cap * 2 : 64;
                idx->ids = (idmap_ent *)realloc(idx->ids, sizeof(idmap_ent) * cap);
            }
            idx->ids[idx->nids].key = key;
            idx->ids[idx->nids].id = id;
            idx->nids++;
        }
        fclose(f);
        qsort(idx->ids, idx->nids, sizeof(idmap_ent), cmp_idmap);
    }
    /* genome chunk lists (readGenomeChunksLists, lib-index-build.go:2193-2245) */
    snprintf(p, sizeof p, "%s/genomes.chunks.bin", dir);
    f = fopen(p, "rb");
    if (f) {
        uint8_t b[8];
        int cap = 0, list = 0;
        while (fread(b, 1, 8, f) == 8) {
            uint64_t n = 0;
            for (int i = 0; i < 8; i++) n = (n << 8) | b[i];
            for (uint64_t j = 0; j < n; j++) {
                if (fread(b, 1, 8, f) != 8) break;
                uint64_t key = 0;
                for (int i = 0; i < 8; i++) key = (key << 8) | b[i];
                if (idx->nchunk_ent == cap) {
                    cap = cap ? cap * 2 : 16;
                    idx->chunk_ents = (struct chunk_ent *)realloc(idx->chunk_ents, sizeof(struct chunk_ent) * cap);
                }
                idx->chunk_ents[idx->nchunk_ent].key = key;
                idx->chunk_ents[idx->nchunk_ent].list = list;
                idx->chunk_ents[idx->nchunk_ent].n = (int)n;
                idx->chunk_ents[idx->nchunk_ent].idx = (int)j;
                idx->nchunk_ent++;
            }
            list++;
        }
        fclose(f);
        idx->has_chunks = idx->nchunk_ent > 0;
        if (idx->has_chunks) qsort(idx->chunk_ents, idx->nchunk_ent, sizeof(struct chunk_ent), cmp_chunk_ent);
    }
    /* SeqComparatorOptions, search.go:360-382 */
    idx->cmpopt.k = 31;
    idx->cmpopt.min_prefix = 11;
    idx->cmpopt.c2.max_gap = opt->align_max_gap;
    idx->cmpopt.c2.min_score = (int)((double)opt->align_min_match_len * opt->align_min_pident / 100);
    idx->cmpopt.c2.min_align_len = opt->align_min_match_len;
    idx->cmpopt.c2.min_identity = opt->align_min_pident;
    idx->cmpopt.c2.band_base = opt->align_band;
    idx->cmpopt.c2.band_count = opt->align_band / 2;
    idx->cmpopt.c2.heuristic_pident = 15;
    idx->cmpopt.min_aligned_fraction = opt->min_qcov_hsp;
    idx->cmpopt.min_identity = opt->align_min_pident;
    idx->chain_min_score = lmo_seed_weight((float)(uint8_t)opt->min_single_prefix); /* :742 */
    return idx;
}

void lmo_index_close(lmo_index *idx) {
    if (!idx) return;
    for (int i = 0; i < idx->nchunks; i++) lmo_kv_free(idx->chunks[i]);
    free(idx->chunks);
    if (idx->grdr)
        for (int i = 0; i < idx->ngbatches; i++) lmo_greader_close(idx->grdr[i]);
    free(idx->grdr);
    for (int i = 0; i < idx->nids; i++) free(idx->ids[i].id);
    free(idx->ids);
    lmo_lh_free(idx->lh);
    free(idx->dir);
    free(idx);
}

static const char *genome_id(const lmo_index *idx, uint64_t key) {
    int lo = 0, hi = idx->nids;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (idx->ids[mid].key < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    if (lo < idx->nids && idx->ids[lo].key == key) return idx->ids[lo].id;
    return "";
}

/* ---- stage 1+2: mask + low-complexity zeroing (lib-index-search.go:1208-1238) ---- */
int lmo_stage_mask(lmo_index *idx, const uint8_t *seq, int len, uint64_t *kmers, int **loc_off, int **locs) {
    if (lmo_lh_mask(idx->lh, seq, len, NULL, 0, 1, kmers, loc_off, locs) != 0) return -1;
    for (int i = 0; i < idx->M; i++)
        if (kmers[i] != 0 && lmo_low_complexity(kmers[i], idx->k)) kmers[i] = 0;
    return 0;
}

static int cmp_anchor(const void *pa, const void *pb) {
    const lmo_anchor *a = (const lmo_anchor *)pa, *b = (const lmo_anchor *)pb;
    if (a->genome != b->genome) return a->genome < b->genome ? -1 : 1;
    if (a->sub.qbegin != b->sub.qbegin) return a->sub.qbegin < b->sub.qbegin ? -1 : 1;
    int ae = a->sub.qbegin + a->sub.len, be = b->sub.qbegin + b->sub.len;
    if (ae != be) return be < ae ? -1 : 1;
    if (a->sub.tbegin != b->sub.tbegin) return a->sub.tbegin < b->sub.tbegin ? -1 : 1;
    if (a->sub.qrc != b->sub.qrc) return a->sub.qrc < b->sub.qrc ? -1 : 1;
    if (a->sub.trc != b->sub.trc) return a->sub.trc < b->sub.trc ? -1 : 1;
    return 0;
}

/* ---- stage 2a/2b/2c: reverse re-bucketing, seed lookup, anchor assembly (:1268-1569) ---- */
int64_t lmo_stage_anchors(lmo_index *idx, const uint64_t *kmers, const int *loc_off, const int *locs,
                          lmo_anchor **out) {
    int M = idx->M, K = idx->k;
    /* 2a: reversed k-mers re-bucketed to argmin mask, de-duplicated per target mask (:1275-1350).
     * Arrival order in the reference is goroutine-dependent; ascending source-mask order is used here. */
    int *tgt = (int *)malloc(sizeof(int) * M);
    uint64_t *rev = (uint64_t *)malloc(sizeof(uint64_t) * M);
    int *cnt = (int *)calloc(M + 1, sizeof(int));
    for (int i = 0; i < M; i++) {
        tgt[i] = -1;
        if (kmers[i] == 0) continue;
        rev[i] = lmo_kmer_reverse(kmers[i], K);
        tgt[i] = lmo_lh_mask_kmer_argmin(idx->lh, rev[i]);
    }
    /* per target mask list with dedup */
    uint64_t *kr = (uint64_t *)malloc(sizeof(uint64_t) * M);
    int *krsrc = (int *)malloc(sizeof(int) * M);
    int *kroff = (int *)calloc(M + 1, sizeof(int));
    for (int i = 0; i < M; i++)
        if (tgt[i] >= 0) cnt[tgt[i] + 1]++;
    for (int i = 0; i < M; i++) cnt[i + 1] += cnt[i];
    int *fill = (int *)calloc(M, sizeof(int));
    uint64_t *tmpk = (uint64_t *)malloc(sizeof(uint64_t) * M);
    int *tmps = (int *)malloc(sizeof(int) * M);
    for (int i = 0; i < M; i++) {
        int t = tgt[i];
        if (t < 0) continue;
        int base = cnt[t], existed = 0;
        for (int j = 0; j < fill[t]; j++)
            if (tmpk[base + j] == rev[i]) {
                existed = 1;
                break;
            }
        if (!existed) {
            tmpk[base + fill[t]] = rev[i];
            tmps[base + fill[t]] = i;
            fill[t]++;
        }
    }
    int nk = 0;
    for (int t = 0; t < M; t++) {
        kroff[t] = nk;
        for (int j = 0; j < fill[t]; j++) {
            kr[nk] = tmpk[cnt[t] + j];
            krsrc[nk] = tmps[cnt[t] + j];
            nk++;
        }
    }
    kroff[M] = nk;
    free(tmpk);
    free(tmps);
    free(fill);
    free(cnt);
    free(tgt);
    free(rev);

    lmo_anchor *anc = NULL;
    int64_t na = 0, capa = 0;
    for (int c = 0; c < idx->nchunks; c++) {
        const lmo_kv_mem *m = idx->chunks[c];
        int begin = m->chunk_index;
        lmo_kv_results res;
        memset(&res, 0, sizeof res);
        int n1;
        lmo_kv_search(m, kmers + begin, idx->opt.min_prefix, 1, 0, &res);
        n1 = res.n;
        /* Search2 wants offsets relative to the chunk */
        int *off2 = (int *)malloc(sizeof(int) * (m->chunk_size + 1));
        for (int i = 0; i <= m->chunk_size; i++) off2[i] = kroff[begin + i] - kroff[begin];
        lmo_kv_search2(m, kr + kroff[begin], off2, idx->opt.min_prefix, 1, 1, &res);
        free(off2);
        (void)n1;
        for (int r = 0; r < res.n; r++) {
            const lmo_kv_sr *sr = &res.sr[r];
            int kprefix = sr->len;
            int srcmask = sr->is_suffix ? krsrc[kroff[sr->iquery] + sr->iquery2] : sr->iquery;
            for (int li = loc_off[srcmask]; li < loc_off[srcmask + 1]; li++) {
                int posq = locs[li];
                int rcq = (posq & 1) > 0;
                posq >>= 1;
                for (int vi = 0; vi < sr->nvals; vi++) {
                    uint64_t refpos = res.vals[sr->val_off + vi];
                    uint64_t bg = refpos >> LMO_BITS_NONE_IDX;
                    int post = (int)((refpos << LMO_BITS_IDX) >> LMO_BITS_IDX_FLAGS);
                    int rvt = (refpos & 1) > 0;
                    int rct = ((refpos >> 1) & 1) > 0;
                    int beginq, begint;
                    if (!rvt) {
                        beginq = rcq ? posq + K - kprefix : posq;
                        begint = rct ? post + K - kprefix : post;
                    } else {
                        beginq = rcq ? posq : posq + K - kprefix;
                        begint = rct ? post : post + K - kprefix;
                    }
                    if (na == capa) {
                        capa = capa ? capa * 2 : 1024;
                        anc = (lmo_anchor *)realloc(anc, sizeof(lmo_anchor) * capa);
                    }
                    lmo_anchor *a = &anc[na++];
                    a->genome = bg;
                    a->sub.qbegin = beginq;
                    a->sub.tbegin = begint;
                    a->sub.len = (uint8_t)kprefix;
                    a->sub.qrc = (uint8_t)rcq;
                    a->sub.trc = (uint8_t)rct;
                    a->sub._pad = 0;
                }
            }
        }
        lmo_kv_results_free(&res);
    }
    free(kr);
    free(krsrc);
    free(kroff);
    if (na > 1) qsort(anc, na, sizeof(lmo_anchor), cmp_anchor);
    *out = anc;
    return na;
}

/* ---------------------------------------------------------------------------------------------- */
static uint8_t rc_tab(uint8_t c) {
    switch (c) {
    case 'A': return 'T';
    case 'C': return 'G';
    case 'G': return 'C';
    case 'T': return 'A';
    case 'a': return 't';
    case 'c': return 'g';
    case 'g': return 'c';
    case 't': return 'a';
    default: return c; /* the genome store only yields ACGT */
    }
}
static void revcomp_inplace(uint8_t *s, int n) { /* RC :2943-2952 */
    for (int i = 0; i < n; i++) s[i] = rc_tab(s[i]);
    for (int i = 0, j = n - 1; i < j; i++, j--) {
        uint8_t t = s[i];
        s[i] = s[j];
        s[j] = t;
    }
}

typedef struct {
    int rc, nseeds;
    double similarity_score;
    lmo_chain2 *chains;
    int nchains;
    int seq_idx, nseqs, seq_len;
    int nchunks, chunk_idx; /* :1118-1119, set at :2375-2385 / :2643-2653 */
    char *seq_id;
} simdetail; /* SimilarityDetail :1099-1120 */

typedef struct {
    uint64_t bg;
    int genome_batch, genome_index;
    lmo_sub *subs;
    int nsubs;
    float score;
    int *chain_off, *chain_idx;
    int nchains;
    int genome_size, num_seqs;
    simdetail *sds;
    int nsds, capsds;
    double aligned_fraction;
    int alive;
} sresult; /* SearchResult :1023-1040 */

typedef struct {
    int qb, qe, tb, te, seq, rc;
} akey;

static void free_chain2(lmo_chain2 *c) {
    free(c->cigar);
    free(c->qseq);
    free(c->tseq);
    free(c->align);
}

static char *fmt_cigar(const lmo_wfa_result *r) { /* :2327-2340: trimmed ops, I<->D swapped */
    int start = -1, end = -1;
    for (int i = 0; i < r->nops; i++)
        if ((r->ops[i] >> 32) == 'M') {
            start = i;
            break;
        }
    for (int i = r->nops - 1; i >= 0; i--)
        if ((r->ops[i] >> 32) == 'M') {
            end = i;
            break;
        }
    char *out = (char *)malloc(16 * (size_t)(r->nops + 1));
    out[0] = 0;
    if (start < 0) return out;
    char *p = out;
    for (int i = start; i <= end; i++) {
        char op = (char)(r->ops[i] >> 32);
        if (op == 'D')
            op = 'I';
        else if (op == 'I')
            op = 'D';
        p += sprintf(p, "%u%c", (unsigned)(r->ops[i] & 0xffffffffu), op);
    }
    return out;
}

/* wfa AlignmentText(q,t,true): aligned region only (between first and last M), '-' for gaps, '|' for matches,
 * ' ' otherwise (columns 22-24 of the -a output; consumed by utils 2blast) */
static void fmt_alignment(const lmo_wfa_result *r, const uint8_t *q, const uint8_t *t, char **Q, char **A, char **T) {
    size_t n = r->align_len + 1;
    *Q = (char *)malloc(n);
    *A = (char *)malloc(n);
    *T = (char *)malloc(n);
    size_t o = 0;
    int start = -1, end = -1;
    for (int i = 0; i < r->nops; i++)
        if ((r->ops[i] >> 32) == 'M') {
            if (start < 0) start = i;
            end = i;
        }
    int qp = 0, tp = 0;
    for (int i = 0; i < r->nops; i++) {
        char op = (char)(r->ops[i] >> 32);
        int cnt = (int)(r->ops[i] & 0xffffffffu);
        int in = start >= 0 && i >= start && i <= end;
        for (int j = 0; j < cnt; j++) {
            if (op == 'M' || op == 'X') {
                if (in) {
                    (*Q)[o] = (char)q[qp];
                    (*T)[o] = (char)t[tp];
                    (*A)[o] = op == 'M' ? '|' : ' ';
                    o++;
                }
                qp++;
                tp++;
            } else if (op == 'I') {
                if (in) {
                    (*Q)[o] = '-';
                    (*T)[o] = (char)t[tp];
                    (*A)[o] = ' ';
                    o++;
                }
                tp++;
            } else if (op == 'D') {
                if (in) {
                    (*Q)[o] = (char)q[qp];
                    (*T)[o] = '-';
                    (*A)[o] = ' ';
                    o++;
                }
                qp++;
            }
        }
    }
    (*Q)[o] = (*A)[o] = (*T)[o] = 0;
}

/* the HSP finalisation block; `variant_a` selects the contig-switch copy (:2223-2357) vs the normal copy
 * (:2490-2626), which differ only in the rc-strand TEnd fix-up (:2285 vs :2552). Returns has_result. */
static int finalize_chains(lmo_index *idx, const uint8_t *s, int qlen, lmo_genome *tseq, int tbegin_w, int tend_w,
                           int rc, lmo_chain2 *chains, int n, int variant_a, double *max_sim_out) {
    const lmo_search_opt *o = &idx->opt;
    int has_result = 0;
    double max_sim = 0;
    for (int i = 0; i < n; i++) {
        lmo_chain2 *c = &chains[i];
        c->aligned_fraction = (double)c->aligned_bases_q / (double)qlen * 100; /* Update2 :240 */
    }
    for (int i = 0; i < n; i++) {
        lmo_chain2 *c = &chains[i];
        if (c->qbegin >= c->qend + 1) {
            c->alive = 0;
            continue;
        }
        int start, end;
        if (rc) {
            start = tend_w - c->tend - c->tpos_offset_begin;
            end = tend_w - c->tbegin - c->tpos_offset_begin + 1;
        } else {
            start = c->tpos_offset_begin + c->tbegin - tbegin_w;
            end = c->tpos_offset_begin + c->tend - tbegin_w + 1;
        }
        if (start >= end) {
            c->alive = 0;
            continue;
        }
        int ext2 = o->ext_len2;
        if (c->aligned_bases_q > 1000000)
            ext2 += 80;
        else if (c->aligned_bases_q > 250000)
            ext2 += 40;
        else if (c->aligned_bases_q > 50000)
            ext2 += 20;
        else if (c->aligned_bases_q > 10000)
            ext2 += 10;
        int s1, e1, s2, e2, qs, qe, ts, te;
        lmo_extend_match(s, qlen, tseq->seq, tseq->seqlen, c->qbegin, c->qend + 1, start, end, ext2, c->tbegin,
                         c->max_ext_len, rc, &qs, &qe, &ts, &te, &s1, &e1, &s2, &e2);
        const uint8_t *_q = s + qs, *_t = tseq->seq + ts;
        int lq = qe - qs, lt = te - ts;
        lmo_wfa_result cg;
        if (lmo_wfa_align(_q, lq, _t, lt, 1, &cg) != 0) {
            c->alive = 0;
            continue;
        }
        lmo_score_evalue(&cg, lq, idx->total_bases, &c->score, &c->bitscore, &c->evalue);
        if (c->evalue > o->max_evalue) {
            c->alive = 0;
            lmo_wfa_result_free(&cg);
            continue;
        }
        c->qbegin -= s1;
        c->qend += e1;
        c->qbegin = c->qbegin + cg.qbegin - 1;
        c->qend = c->qend - (lq - cg.qend);
        if (rc) {
            c->tbegin -= e2;
            c->tend += s2;
            c->tbegin = c->tbegin + (lt - cg.tend);
            if (variant_a)
                c->tend = c->tend - cg.tbegin - 1; /* :2285, kept as written */
            else
                c->tend = c->tend - (cg.tbegin - 1); /* :2552 */
        } else {
            c->tbegin -= s2;
            c->tend += e2;
            c->tbegin = c->tbegin + cg.tbegin - 1;
            c->tend = c->tend - (lt - cg.tend);
        }
        c->aligned_bases_q = c->qend - c->qbegin + 1;
        c->aligned_length = (int)cg.align_len;
        c->matched_bases = (int)cg.matches;
        c->gaps = (int)cg.gaps;
        c->aligned_fraction = (double)c->aligned_bases_q / (double)qlen * 100;
        if (c->aligned_fraction > 100) c->aligned_fraction = 100;
        c->pident = (double)c->matched_bases / (double)cg.align_len * 100;
        if (c->aligned_fraction < idx->cmpopt.min_aligned_fraction || c->pident < idx->cmpopt.min_identity) {
            c->alive = 0;
            lmo_wfa_result_free(&cg);
            continue;
        }
        if (o->output_seq) {
            c->cigar = fmt_cigar(&cg);
            fmt_alignment(&cg, _q, _t, &c->qseq, &c->align, &c->tseq);
        }
        lmo_wfa_result_free(&cg);
        double sim = (double)c->bitscore * c->pident;
        if (sim > max_sim) max_sim = sim;
        has_result = 1;
    }
    *max_sim_out = max_sim;
    return has_result;
}

/* aligned bases of the query over all HSPs of the genome (:2701-2738, again at :2855-2890); 0 = filtered out */
static int genome_qcov_filter(sresult *r, int qlen, double min_qcov_genome) {
    int nreg = 0;
    for (int i = 0; i < r->nsds; i++)
        for (int j = 0; j < r->sds[i].nchains; j++)
            if (r->sds[i].chains[j].alive) nreg++;
    int(*regions)[2] = (int(*)[2])malloc(sizeof(int[2]) * (nreg ? nreg : 1));
    nreg = 0;
    for (int i = 0; i < r->nsds; i++)
        for (int j = 0; j < r->sds[i].nchains; j++)
            if (r->sds[i].chains[j].alive) {
                regions[nreg][0] = r->sds[i].chains[j].qbegin;
                regions[nreg][1] = r->sds[i].chains[j].qend;
                nreg++;
            }
    int ab = lmo_coverage_len(regions, nreg);
    free(regions);
    r->aligned_fraction = (double)ab / (double)qlen * 100;
    if (r->aligned_fraction > 100) r->aligned_fraction = 100;
    return r->aligned_fraction >= min_qcov_genome;
}
/* HSP clusters by SimilarityScore descending (:2745, :2895), stable */
static void sort_sds(sresult *r) {
    for (int i = 1; i < r->nsds; i++) {
        simdetail x = r->sds[i];
        int j = i - 1;
        while (j >= 0 && r->sds[j].similarity_score < x.similarity_score) {
            r->sds[j + 1] = r->sds[j];
            j--;
        }
        r->sds[j + 1] = x;
    }
}

static void add_sd(const lmo_index *idx, sresult *r, lmo_genome *tseq, int iseq, int rc, int nseeds, double max_sim,
                   lmo_chain2 *chains, int n) {
    if (r->nsds == r->capsds) {
        r->capsds = r->capsds ? r->capsds * 2 : 4;
        r->sds = (simdetail *)realloc(r->sds, sizeof(simdetail) * r->capsds);
    }
    simdetail *sd = &r->sds[r->nsds++];
    sd->rc = rc;
    sd->nseeds = nseeds;
    sd->similarity_score = max_sim;
    sd->chains = chains;
    sd->nchains = n;
    sd->seq_idx = iseq;
    sd->nseqs = tseq->nseqs;
    sd->seq_len = tseq->seq_sizes[iseq];
    sd->seq_id = strdup(tseq->seq_ids[iseq]);
    sd->nchunks = 1;
    sd->chunk_idx = 0;
    if (idx->has_chunks) {
        const struct chunk_ent *ce = find_chunk(idx, r->bg);
        if (ce) {
            sd->nchunks = ce->n;
            sd->chunk_idx = ce->idx;
        }
    }
}

static int akey_seen(akey **keys, int *nk, int *capk, akey k) {
    for (int i = 0; i < *nk; i++) {
        akey *x = &(*keys)[i];
        if (x->qb == k.qb && x->qe == k.qe && x->tb == k.tb && x->te == k.te && x->seq == k.seq && x->rc == k.rc)
            return 1;
    }
    if (*nk == *capk) {
        *capk = *capk ? *capk * 2 : 16;
        *keys = (akey *)realloc(*keys, sizeof(akey) * *capk);
    }
    (*keys)[(*nk)++] = k;
    return 0;
}

/* falin, lib-index-search.go:1887-2763 */
static void align_genome(lmo_index *idx, const uint8_t *s, int qlen, lmo_cmp *cpr, sresult *r) {
    const lmo_search_opt *o = &idx->opt;
    int K = idx->k;
    int ext_len = o->ext_len, contig_interval = idx->contig_interval;
    lmo_greader *rdr = idx->grdr[r->genome_batch];
    lmo_genome *tseq = NULL;
    akey *keys = NULL;
    int nkeys = 0, capkeys = 0;
    /* sort chains by TBegin of their first anchor (:1967-1974), stable */
    int nch = r->nchains;
    int *order = (int *)malloc(sizeof(int) * (nch ? nch : 1));
    for (int i = 0; i < nch; i++) order[i] = i;
    for (int i = 1; i < nch; i++) {
        int x = order[i];
        int32_t tx = r->subs[r->chain_idx[r->chain_off[x]]].tbegin;
        int j = i - 1;
        while (j >= 0 && r->subs[r->chain_idx[r->chain_off[order[j]]]].tbegin > tx) {
            order[j + 1] = order[j];
            j--;
        }
        order[j + 1] = x;
    }
    for (int ci = 0; ci < nch; ci++) {
        const int *chain = r->chain_idx + r->chain_off[order[ci]];
        int nseeds = r->chain_off[order[ci] + 1] - r->chain_off[order[ci]];
        const lmo_sub *sub = &r->subs[chain[0]];
        int qb = sub->qbegin, tb = sub->tbegin;
        sub = &r->subs[chain[nseeds - 1]];
        int qe = sub->qbegin + sub->len - 1, te = sub->tbegin + sub->len - 1;
        int rc;
        if (nseeds == 1)
            rc = sub->qrc != sub->trc;
        else
            rc = tb > sub->tbegin;
        int tBegin, tEnd;
        if (rc) {
            tBegin = sub->tbegin - ext_len;
            if (tBegin < 0) tBegin = 0;
            tEnd = tb + sub->len - 1 + ext_len;
        } else {
            tBegin = tb - ext_len;
            if (tBegin < 0) tBegin = 0;
            tEnd = te + ext_len;
        }
        int qBegin = qb - (qb < ext_len ? qb : ext_len);
        int qEnd = qe + (qlen - qe - 1 < ext_len ? qlen - qe - 1 : ext_len);
        tseq = lmo_subseq3(rdr, r->genome_index, tBegin, tEnd, tseq);
        if (!tseq) break;
        if (tseq->seqlen < tEnd - tBegin + 1) tEnd -= tEnd - tBegin + 1 - tseq->seqlen;
        if (rc) revcomp_inplace(tseq->seq, tseq->seqlen);
        lmo_chain2 *crchains = NULL;
        int ncr = lmo_cmp_compare(cpr, (uint32_t)qBegin, (uint32_t)qEnd, tseq->seq, tseq->seqlen, qlen, &crchains,
                                  NULL, NULL);
        if (ncr == 0) continue;
        if (r->genome_size == 0) {
            r->genome_size = tseq->genome_size;
            r->num_seqs = tseq->nseqs;
        }
        int iSeqPre = -1, iSeq = 0;
        int tPosOffsetBegin = 0, tPosOffsetEnd = 0;
        lmo_chain2 *cur = (lmo_chain2 *)malloc(sizeof(lmo_chain2) * ncr);
        int ncur = 0;
        int seqlen_w = tseq->seqlen;
        for (int _i = 0; _i < ncr; _i++) {
            lmo_chain2 *c = &crchains[_i];
            qb = c->qbegin;
            qe = c->qend;
            tb = c->tbegin;
            te = c->tend;
            iSeq = 0;
            tPosOffsetBegin = 0;
            tPosOffsetEnd = 0;
            if (tseq->nseqs > 1) {
                iSeq = -1;
                int _begin, _end;
                if (rc) {
                    _begin = tEnd - te + K;
                    _end = tEnd - tb - K;
                } else {
                    _begin = tBegin + tb + K;
                    _end = tBegin + te - K;
                }
                if (_begin >= _end) {
                    if (rc) {
                        _begin = tEnd - te;
                        _end = tEnd - tb;
                    } else {
                        _begin = tBegin + tb;
                        _end = tBegin + te;
                    }
                }
                for (int j = 0; j < tseq->nseqs; j++) {
                    int l = tseq->seq_sizes[j];
                    tPosOffsetEnd += l - 1;
                    if (_begin + K >= tPosOffsetBegin && _end - K <= tPosOffsetEnd) {
                        iSeq = j;
                        break;
                    } else if (_end < tPosOffsetBegin) {
                        iSeq = -1;
                        break;
                    }
                    tPosOffsetEnd += contig_interval + 1;
                    tPosOffsetBegin = tPosOffsetEnd;
                }
                if (iSeq < 0) {
                    c->alive = 0;
                    continue;
                }
                if (iSeqPre >= 0 && iSeq != iSeqPre) {
                    int iSeq0 = iSeq;
                    iSeq = iSeqPre;
                    /* convert positions (:2167-2200) */
                    c->qbegin = qb;
                    c->qend = qe;
                    c->tpos_offset_begin = tPosOffsetBegin;
                    if (rc) {
                        c->tbegin = tBegin - tPosOffsetBegin + (seqlen_w - te - 1);
                        if (c->tbegin < 0) {
                            c->qend += c->tbegin;
                            c->aligned_bases_q += c->tbegin;
                            c->tbegin = 0;
                        }
                        c->tend = tBegin - tPosOffsetBegin + (seqlen_w - tb - 1);
                        if (c->tend > tseq->seq_sizes[iSeq] - 1) {
                            c->qbegin += c->tend - (tseq->seq_sizes[iSeq] - 1);
                            c->tend = tseq->seq_sizes[iSeq] - 1;
                        }
                    } else {
                        c->tbegin = tBegin - tPosOffsetBegin + tb;
                        if (c->tbegin < 0) {
                            c->qbegin -= c->tbegin;
                            c->aligned_bases_q += c->tbegin;
                            c->tbegin = 0;
                        }
                        c->tend = tBegin - tPosOffsetBegin + te;
                        if (c->tend > tseq->seq_sizes[iSeq] - 1) {
                            c->qend -= c->tend - (tseq->seq_sizes[iSeq] - 1);
                            c->tend = tseq->seq_sizes[iSeq] - 1;
                        }
                    }
                    c->max_ext_len = tseq->seq_sizes[iSeq] - 1 - c->tend;
                    if (ncur > 0) {
                        double max_sim;
                        int has = finalize_chains(idx, s, qlen, tseq, tBegin, tEnd, rc, cur, ncur, 1, &max_sim);
                        if (has) {
                            add_sd(idx, r, tseq, iSeq, rc, nseeds, max_sim, cur, ncur);
                        } else {
                            for (int z = 0; z < ncur; z++) free_chain2(&cur[z]);
                            free(cur);
                        }
                        cur = (lmo_chain2 *)malloc(sizeof(lmo_chain2) * ncr);
                        ncur = 0;
                    }
                    iSeqPre = -1;
                    /* (the reference allocates a fresh crChains2 here even when the previous one was empty) */
                    akey key = {c->qbegin, c->qend, c->tbegin, c->tend, iSeq, rc};
                    if (akey_seen(&keys, &nkeys, &capkeys, key)) {
                        c->alive = 0;
                    } else {
                        cur[ncur++] = *c;
                    }
                    iSeq = iSeq0;
                    continue;
                }
            }
            iSeqPre = iSeq;
            c->qbegin = qb;
            c->qend = qe;
            c->tpos_offset_begin = tPosOffsetBegin;
            if (rc) {
                c->tbegin = tBegin - tPosOffsetBegin + (seqlen_w - te - 1);
                if (c->tbegin < 0) {
                    c->qend += c->tbegin;
                    c->aligned_bases_q += c->tbegin;
                    c->tbegin = 0;
                }
                c->tend = tBegin - tPosOffsetBegin + (seqlen_w - tb - 1);
                if (c->tend > tseq->seq_sizes[iSeq] - 1) {
                    c->qbegin += c->tend - (tseq->seq_sizes[iSeq] - 1);
                    c->tend = tseq->seq_sizes[iSeq] - 1;
                }
            } else {
                c->tbegin = tBegin - tPosOffsetBegin + tb;
                if (c->tbegin < 0) {
                    c->qbegin -= c->tbegin;
                    c->aligned_bases_q += c->tbegin;
                    c->tbegin = 0;
                }
                c->tend = tBegin - tPosOffsetBegin + te;
                if (c->tend > tseq->seq_sizes[iSeq] - 1) {
                    c->qend -= c->tend - (tseq->seq_sizes[iSeq] - 1);
                    c->tend = tseq->seq_sizes[iSeq] - 1;
                }
            }
            c->max_ext_len = tseq->seq_sizes[iSeq] - 1 - c->tend;
            akey key = {c->qbegin, c->qend, c->tbegin, c->tend, iSeq, rc};
            if (akey_seen(&keys, &nkeys, &capkeys, key)) {
                c->alive = 0;
            } else {
                cur[ncur++] = *c;
            }
        }
        int used = 0;
        if (iSeq >= 0) {
            if (ncur > 0) {
                double max_sim;
                int has = finalize_chains(idx, s, qlen, tseq, tBegin, tEnd, rc, cur, ncur, 0, &max_sim);
                if (has) {
                    add_sd(idx, r, tseq, iSeq, rc, nseeds, max_sim, cur, ncur);
                    used = 1;
                }
            }
        }
        if (!used) {
            for (int z = 0; z < ncur; z++) free_chain2(&cur[z]);
            free(cur);
        }
        free(crchains);
    }
    free(order);
    free(keys);
    lmo_genome_free(tseq);
    if (r->nsds == 0) {
        r->alive = 0;
        return;
    }
    /* query coverage per genome (:2701-2738); with genome chunks in the index it is computed after the chunk merge */
    if (!idx->has_chunks && !genome_qcov_filter(r, qlen, o->min_qcov_genome)) {
        r->alive = 0;
        return;
    }
    sort_sds(r);
}

/* SortBySeqID :1042-1096: regroup clusters by sseqid preserving first-seen order */
static void sort_by_seqid(sresult *r) {
    if (r->nsds <= 1) return;
    simdetail *out = (simdetail *)malloc(sizeof(simdetail) * r->nsds);
    uint8_t *used = (uint8_t *)calloc(r->nsds, 1);
    int n = 0;
    for (int i = 0; i < r->nsds; i++) {
        if (used[i]) continue;
        for (int j = i; j < r->nsds; j++) {
            if (!used[j] && strcmp(r->sds[j].seq_id, r->sds[i].seq_id) == 0) {
                out[n++] = r->sds[j];
                used[j] = 1;
            }
        }
    }
    memcpy(r->sds, out, sizeof(simdetail) * r->nsds);
    free(out);
    free(used);
}

static void free_sresult(sresult *r) {
    free(r->subs);
    free(r->chain_off);
    free(r->chain_idx);
    for (int i = 0; i < r->nsds; i++) {
        for (int j = 0; j < r->sds[i].nchains; j++) free_chain2(&r->sds[i].chains[j]);
        free(r->sds[i].chains);
        free(r->sds[i].seq_id);
    }
    free(r->sds);
}

static int cmp_score_desc(const void *a, const void *b) {
    const sresult *x = (const sresult *)a, *y = (const sresult *)b;
    if (x->score != y->score) return x->score > y->score ? -1 : 1;
    return x->bg < y->bg ? -1 : x->bg > y->bg;
}
static int cmp_genome_index(const void *a, const void *b) {
    const sresult *x = (const sresult *)a, *y = (const sresult *)b;
    if (x->genome_index != y->genome_index) return x->genome_index < y->genome_index ? -1 : 1;
    return x->bg < y->bg ? -1 : x->bg > y->bg;
}
static int cmp_best_sim(const void *a, const void *b) {
    const sresult *x = (const sresult *)a, *y = (const sresult *)b;
    double sx = x->sds[0].similarity_score, sy = y->sds[0].similarity_score;
    if (sx != sy) return sx > sy ? -1 : 1;
    return x->bg < y->bg ? -1 : x->bg > y->bg;
}

void lmo_result_free(lmo_result *r) {
    for (int i = 0; i < r->n; i++) {
        free(r->rows[i].cigar);
        free(r->rows[i].qseq);
        free(r->rows[i].tseq);
        free(r->rows[i].align);
        free((char *)r->rows[i].seq_id);
    }
    free(r->rows);
    memset(r, 0, sizeof *r);
}

int lmo_search(lmo_index *idx, const uint8_t *seq, int len, lmo_result *res) {
    memset(res, 0, sizeof *res);
    int M = idx->M, K = idx->k;
    if (len < K) return 0;
    uint64_t *kmers = (uint64_t *)malloc(sizeof(uint64_t) * M);
    int *loc_off = NULL, *locs = NULL;
    if (lmo_stage_mask(idx, seq, len, kmers, &loc_off, &locs) != 0) {
        free(kmers);
        return -1;
    }
    lmo_anchor *anc = NULL;
    int64_t na = lmo_stage_anchors(idx, kmers, loc_off, locs, &anc);
    free(kmers);
    free(loc_off);
    free(locs);
    res->n_anchors = na;
    if (na == 0) {
        free(anc);
        return 0;
    }
    /* 3.1 chaining per genome (:1702-1775) */
    sresult *rs = NULL;
    int nrs = 0, caprs = 0;
    for (int64_t i = 0; i < na;) {
        int64_t j = i;
        while (j < na && anc[j].genome == anc[i].genome) j++;
        res->n_genomes_seeded++;
        int n = (int)(j - i);
        lmo_sub *subs = (lmo_sub *)malloc(sizeof(lmo_sub) * n);
        for (int t = 0; t < n; t++) subs[t] = anc[i + t].sub;
        if (n > 1) n = lmo_clear_subs(subs, n, K);
        int *coff, *cidx, nch;
        float score = lmo_chainer(subs, n, (float)idx->opt.max_gap, idx->chain_min_score,
                                  (float)idx->opt.max_distance, idx->opt.top_n_chains, &coff, &cidx, &nch);
        if (score < idx->chain_min_score) {
            free(subs);
            free(coff);
            free(cidx);
        } else {
            if (nrs == caprs) {
                caprs = caprs ? caprs * 2 : 16;
                rs = (sresult *)realloc(rs, sizeof(sresult) * caprs);
            }
            sresult *r = &rs[nrs++];
            memset(r, 0, sizeof *r);
            r->bg = anc[i].genome;
            r->genome_batch = (int)(r->bg >> LMO_BITS_GENOME_IDX);
            r->genome_index = (int)(r->bg & LMO_MASK_GENOME_IDX);
            r->subs = subs;
            r->nsubs = n;
            r->score = score;
            r->chain_off = coff;
            r->chain_idx = cidx;
            r->nchains = nch;
            r->alive = 1;
            res->n_chains += nch;
        }
        i = j;
    }
    free(anc);
    /* 3.2 top N (:1781-1805) */
    if (idx->opt.top_n > 0 && nrs > idx->opt.top_n) {
        qsort(rs, nrs, sizeof(sresult), cmp_score_desc);
        for (int i = idx->opt.top_n; i < nrs; i++) free_sresult(&rs[i]);
        nrs = idx->opt.top_n;
    }
    if (nrs == 0) {
        free(rs);
        return 0;
    }
    /* 3.3 alignment */
    lmo_cmp *cpr = lmo_cmp_new(&idx->cmpopt);
    lmo_cmp_index(cpr, seq, len);
    if (nrs > 1) qsort(rs, nrs, sizeof(sresult), cmp_genome_index);
    for (int i = 0; i < nrs; i++) align_genome(idx, seq, len, cpr, &rs[i]);
    lmo_cmp_free(cpr);
    int n2 = 0;
    for (int i = 0; i < nrs; i++) {
        if (rs[i].alive)
            rs[n2++] = rs[i];
        else
            free_sresult(&rs[i]);
    }
    nrs = n2;
    if (nrs == 0) {
        free(rs);
        return 0;
    }
    /* merge the results of the chunks of one split genome (:2798-2913). The reference merges into whichever chunk comes
     * first in its goroutine-arrival-ordered list; here the list is in genome order, so the first chunk of the genome. */
    if (idx->has_chunks) {
        for (int i = 0; i < nrs; i++) {
            if (!rs[i].alive) continue;
            const struct chunk_ent *ci = find_chunk(idx, rs[i].bg);
            if (!ci) continue;
            for (int j = i + 1; j < nrs; j++) {
                if (!rs[j].alive) continue;
                const struct chunk_ent *cj = find_chunk(idx, rs[j].bg);
                if (!cj || cj->list != ci->list) continue;
                /* only SimilarityDetails and AlignedFraction need updating (:2832-2840) */
                for (int a = 0; a < rs[j].nsds; a++) {
                    if (rs[i].nsds == rs[i].capsds) {
                        rs[i].capsds = rs[i].capsds ? rs[i].capsds * 2 : 4;
                        rs[i].sds = (simdetail *)realloc(rs[i].sds, sizeof(simdetail) * rs[i].capsds);
                    }
                    rs[i].sds[rs[i].nsds++] = rs[j].sds[a];
                }
                rs[j].nsds = 0;
                rs[j].alive = 0;
            }
        }
        /* recompute the query coverage per genome, filter, re-sort the clusters (:2853-2897): every result, merged or not */
        n2 = 0;
        for (int i = 0; i < nrs; i++) {
            if (rs[i].alive && genome_qcov_filter(&rs[i], len, idx->opt.min_qcov_genome)) {
                sort_sds(&rs[i]);
                rs[n2++] = rs[i];
            } else {
                free_sresult(&rs[i]);
            }
        }
        nrs = n2;
        if (nrs == 0) {
            free(rs);
            return 0;
        }
    }
    qsort(rs, nrs, sizeof(sresult), cmp_best_sim);
    for (int i = 0; i < nrs; i++) sort_by_seqid(&rs[i]);
    /* flatten to rows in printing order (search.go:468-523) */
    res->ngenomes = nrs;
    for (int i = 0; i < nrs; i++) {
        sresult *r = &rs[i];
        int cls = 1, hsp = 1;
        for (int a = 0; a < r->nsds; a++) {
            simdetail *sd = &r->sds[a];
            for (int b = 0; b < sd->nchains; b++) {
                lmo_chain2 *c = &sd->chains[b];
                if (!c->alive) continue;
                if (res->n == res->cap) {
                    res->cap = res->cap ? res->cap * 2 : 16;
                    res->rows = (lmo_hsp *)realloc(res->rows, sizeof(lmo_hsp) * res->cap);
                }
                lmo_hsp *h = &res->rows[res->n++];
                memset(h, 0, sizeof *h);
                h->batch_genome = r->bg;
                h->qcov_genome = r->aligned_fraction;
                h->cls = cls;
                h->hsp = hsp;
                h->seq_idx = sd->seq_idx;
                h->nseqs = sd->nseqs;
                h->seq_len = sd->seq_len;
                h->nchunks = sd->nchunks;
                h->chunk_idx = sd->chunk_idx;
                h->rc = sd->rc;
                h->qcov_hsp = c->aligned_fraction;
                h->aligned_length = c->aligned_length;
                h->pident = c->pident;
                h->gaps = c->gaps;
                h->qbegin = c->qbegin;
                h->qend = c->qend;
                h->tbegin = c->tbegin;
                h->tend = c->tend;
                h->evalue = c->evalue;
                h->bitscore = c->bitscore;
                h->score = c->score;
                h->matched_bases = c->matched_bases;
                h->cigar = c->cigar;
                h->qseq = c->qseq;
                h->tseq = c->tseq;
                h->align = c->align;
                c->cigar = c->qseq = c->tseq = c->align = NULL;
                h->genome_id = genome_id(idx, r->bg);
                h->seq_id = strdup(sd->seq_id);
                hsp++;
            }
            cls++;
        }
        free_sresult(r);
    }
    free(rs);
    return 0;
}

/* Go's %.2e prints at least two exponent digits; C's printf does the same ("e+00"). */
int lmo_format_row(const lmo_hsp *h, const char *query_id, int qlen, int hits, int more_columns, char *buf,
                   size_t buflen) {
    int n = snprintf(buf, buflen, "%s\t%d\t%d\t%s\t%s\t%.3f\t%d\t%d\t%.3f\t%d\t%.3f\t%d\t%d\t%d\t%d\t%d\t%c\t%d\t%.2e\t%d",
                     query_id, qlen, hits, h->genome_id, h->seq_id, h->qcov_genome, h->cls, h->hsp, h->qcov_hsp,
                     h->aligned_length, h->pident, h->gaps, h->qbegin + 1, h->qend + 1, h->tbegin + 1, h->tend + 1,
                     h->rc ? '-' : '+', h->seq_len, h->evalue, h->bitscore);
    if (more_columns && n > 0 && (size_t)n < buflen)
        n += snprintf(buf + n, buflen - n, "\t%s\t%s\t%s\t%s", h->cigar ? h->cigar : "", h->qseq ? h->qseq : "",
                      h->tseq ? h->tseq : "", h->align ? h->align : "");
    return n;
}
