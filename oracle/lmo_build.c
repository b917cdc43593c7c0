/* CPU ORACLE (test infrastructure only) — synthetic-index writer.
 * Index BUILDING is out of the hot-path scope; this file exists because no reference-built index
 * can exist here.  It follows the reference builder closely enough to produce indexes of the same shape and in the
 * same on-disk format:
 *   genome concatenation with contig-interval 'A's   lib-index-build.go:924,1662-1678
 *   skip regions (spacers, >=5 N runs)               lib-index-build.go:971-1016, lib-gaps.go:38-60
 *   masking + low-complexity removal                 lib-index-build.go:1028-1046
 *   seed-desert filling                              lib-index-build.go:1094-1407
 *   seed values + reversed (suffix) seeds            lib-index-build.go:642-890
 *   chunk files / info.toml / map / chunks           lib-index-build.go:1855-1889,1914-1932,649-655,1787-1808
 * Not reproduced: genome splitting above --max-genome, --max-kmer-freq, soft-masking, multi-batch merge.
 */
#include "lmo.h"
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <errno.h>

void lmo_build_opt_default(lmo_build_opt *o) {
    o->k = 31;
    o->masks = 20000;     /* index.go:560 */
    o->rand_seed = 1;
    o->max_desert = 100;  /* index.go:582 */
    o->seed_dist = 50;    /* index.go:584 */
    o->chunks = 8;
    o->partitions = 4096; /* index.go:603 */
    o->batch_size = 5000; /* index.go:613 */
    o->contig_interval = 1000; /* index.go:619 */
    o->max_genome = 20000000;  /* index.go:538 */
}

typedef struct {
    uint64_t kmer, val;
} kvpair;
typedef struct {
    kvpair *v;
    int64_t n, cap;
} kvvec;
static void kv_push(kvvec *a, uint64_t kmer, uint64_t val) {
    if (a->n == a->cap) {
        a->cap = a->cap ? a->cap * 2 : 16;
        a->v = (kvpair *)realloc(a->v, sizeof(kvpair) * a->cap);
    }
    a->v[a->n].kmer = kmer;
    a->v[a->n].val = val;
    a->n++;
}

struct lmo_builder {
    char *dir;
    lmo_build_opt opt;
    lmo_lh *lh;
    kvvec *data; /* per mask */
    int ngenomes;
    int64_t input_bases;
    lmo_gwriter *gw;
    int cur_batch;
    FILE *fmap;
    /* genome chunks (lib-index-build.go:1786-1808): lists of batch+index keys of the records of one split genome */
    uint64_t *chunk_keys;
    int *chunk_list_len;
    int nchunk_keys, nchunk_lists, cap_chunk_keys, cap_chunk_lists;
    int ninput;
};

static void mkdir_p(const char *p) {
    if (mkdir(p, 0777) != 0 && errno != EEXIST) { /* ignore */
    }
}

lmo_builder *lmo_builder_new(const char *outdir, const lmo_build_opt *opt) {
    lmo_builder *b = (lmo_builder *)calloc(1, sizeof *b);
    b->dir = strdup(outdir);
    b->opt = *opt;
    uint64_t *masks = (uint64_t *)malloc(sizeof(uint64_t) * opt->masks);
    lmo_gen_masks(opt->k, opt->masks, opt->rand_seed, masks);
    b->lh = lmo_lh_new(opt->k, masks, opt->masks);
    free(masks);
    b->data = (kvvec *)calloc(opt->masks, sizeof(kvvec));
    b->cur_batch = -1;
    char p[4096];
    mkdir_p(outdir);
    snprintf(p, sizeof p, "%s/seeds", outdir);
    mkdir_p(p);
    snprintf(p, sizeof p, "%s/genomes", outdir);
    mkdir_p(p);
    snprintf(p, sizeof p, "%s/masks.bin", outdir);
    lmo_lh_write(b->lh, p, opt->rand_seed);
    snprintf(p, sizeof p, "%s/genomes.map.bin", outdir);
    b->fmap = fopen(p, "wb");
    return b;
}

/* Test hook: replace the generated mask set by a caller-provided one (sorted ascending, every p-base prefix present) before
 * the first genome is added - lets the tests index genomes with the mask set of the GPU synthetic builder and compare
 * what both writers store. masks.bin is rewritten. */
int lmo_builder_set_masks(lmo_builder *b, const uint64_t *masks, int n) {
    if (!b || n != b->opt.masks || b->ngenomes != 0) return -1;
    lmo_lh_free(b->lh);
    b->lh = lmo_lh_new(b->opt.k, masks, n);
    char p[4096];
    snprintf(p, sizeof p, "%s/masks.bin", b->dir);
    lmo_lh_write(b->lh, p, b->opt.rand_seed);
    return 0;
}

typedef struct {
    int s, e;
} ivl;

static int in_intervals(const ivl *iv, int n, int pos) {
    for (int i = 0; i < n; i++)
        if (pos >= iv[i].s && pos <= iv[i].e) return 1;
    return 0;
}

static int cmp_u32(const void *a, const void *b) {
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : x > y;
}
static int cmp_skip(const void *a, const void *b) {
    const int *x = (const int *)a, *y = (const int *)b;
    return x[0] < y[0] ? -1 : x[0] > y[0];
}

static int valid_seed_kmer(uint64_t kmer, int k) { return kmer != 0 && !lmo_low_complexity(kmer, k); }

static int add_one(lmo_builder *b, const char *id, int ncontigs, const char **contig_ids, const uint8_t **contigs,
                   const int *contig_lens, uint64_t *bg_out);

int lmo_builder_add(lmo_builder *b, const char *id, int ncontigs, const char **contig_ids, const uint8_t **contigs,
                    const int *contig_lens) {
    /* lib-index-build.go:1581-1658: contigs are appended while the concatenation (with spacers) stays within max_genome;
     * a contig that would overflow it closes the current chunk and starts the next one */
    const lmo_build_opt *o = &b->opt;
    const int64_t maxg = o->max_genome > 0 ? o->max_genome : ((int64_t)1 << 28) - 1;
    for (int i = 0; i < ncontigs; i++)
        if (contig_lens[i] > maxg) return -2; /* "skipping a big genome with a sequence of %d bp" */
    uint64_t *keys = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(ncontigs + 1));
    int nk = 0, first = 0, rc = 0;
    int64_t cur = 0; /* len(refseq.Seq) */
    for (int i = 0; i < ncontigs && rc == 0; i++) {
        if (cur + contig_lens[i] > maxg && i > first) {
            rc = add_one(b, id, i - first, contig_ids + first, contigs + first, contig_lens + first, &keys[nk]);
            nk++;
            first = i;
            cur = 0;
        }
        if (i > first) cur += o->contig_interval;
        cur += contig_lens[i];
    }
    if (rc == 0 && first < ncontigs) {
        rc = add_one(b, id, ncontigs - first, contig_ids + first, contigs + first, contig_lens + first, &keys[nk]);
        nk++;
    }
    if (rc == 0 && nk > 1) {
        if (b->nchunk_keys + nk > b->cap_chunk_keys) {
            b->cap_chunk_keys = (b->nchunk_keys + nk) * 2;
            b->chunk_keys = (uint64_t *)realloc(b->chunk_keys, sizeof(uint64_t) * (size_t)b->cap_chunk_keys);
        }
        if (b->nchunk_lists == b->cap_chunk_lists) {
            b->cap_chunk_lists = b->cap_chunk_lists ? b->cap_chunk_lists * 2 : 8;
            b->chunk_list_len = (int *)realloc(b->chunk_list_len, sizeof(int) * (size_t)b->cap_chunk_lists);
        }
        memcpy(b->chunk_keys + b->nchunk_keys, keys, sizeof(uint64_t) * (size_t)nk);
        b->nchunk_keys += nk;
        b->chunk_list_len[b->nchunk_lists++] = nk;
    }
    free(keys);
    if (rc == 0) b->ninput++;
    return rc;
}

static int add_one(lmo_builder *b, const char *id, int ncontigs, const char **contig_ids, const uint8_t **contigs,
                   const int *contig_lens, uint64_t *bg_out) {
    const lmo_build_opt *o = &b->opt;
    int k = o->k, M = o->masks, interval = o->contig_interval;
    /* concatenate */
    int64_t total = 0, gsize = 0;
    for (int i = 0; i < ncontigs; i++) {
        total += contig_lens[i];
        gsize += contig_lens[i];
    }
    total += (int64_t)(ncontigs - 1) * interval;
    if (total >= ((int64_t)1 << 28) || total < k) return -1;
    int len = (int)total;
    uint8_t *seq = (uint8_t *)malloc(len + 1);
    int *skip = (int *)malloc(sizeof(int) * 2 * (ncontigs + 16));
    int nskip = 0, capskip = ncontigs + 16;
    ivl *iv = (ivl *)malloc(sizeof(ivl) * (ncontigs + 16));
    int niv = 0, capiv = ncontigs + 16;
    int n = 0;
    for (int i = 0; i < ncontigs; i++) {
        if (i > 0) {
            skip[2 * nskip] = n;
            skip[2 * nskip + 1] = n + interval - 1;
            nskip++;
            iv[niv].s = n - k + 1;
            iv[niv].e = n + interval - 1;
            niv++;
            memset(seq + n, 'A', interval);
            n += interval;
        }
        memcpy(seq + n, contigs[i], contig_lens[i]);
        n += contig_lens[i];
    }
    /* gaps of >= 5 N (lib-gaps.go:38) */
    int had_gaps = 0;
    for (int i = 0; i < len;) {
        if (seq[i] != 'N' && seq[i] != 'n') {
            i++;
            continue;
        }
        int st = i;
        i++;
        while (i < len && (seq[i] == 'N' || seq[i] == 'n')) i++;
        if (i - st >= 5) {
            if (nskip == capskip) {
                capskip *= 2;
                skip = (int *)realloc(skip, sizeof(int) * 2 * capskip);
            }
            if (niv == capiv) {
                capiv *= 2;
                iv = (ivl *)realloc(iv, sizeof(ivl) * capiv);
            }
            skip[2 * nskip] = st;
            skip[2 * nskip + 1] = i - 1;
            nskip++;
            iv[niv].s = st - k + 1;
            iv[niv].e = i - 1;
            niv++;
            had_gaps = 1;
        }
    }
    if (had_gaps) qsort(skip, nskip, sizeof(int) * 2, cmp_skip);

    /* genome record */
    int batch = b->ngenomes / o->batch_size, gidx = b->ngenomes % o->batch_size;
    if (batch != b->cur_batch) {
        if (b->gw) lmo_gwriter_close(b->gw);
        char p[4096];
        snprintf(p, sizeof p, "%s/genomes/batch_%04d", b->dir, batch);
        mkdir_p(p);
        snprintf(p, sizeof p, "%s/genomes/batch_%04d/genomes.bin", b->dir, batch);
        b->gw = lmo_gwriter_open(p, (uint32_t)batch);
        b->cur_batch = batch;
    }
    lmo_genome_in gi;
    gi.id = (char *)id;
    gi.genome_size = (int)gsize;
    gi.len = len;
    gi.nseqs = ncontigs;
    gi.seq_sizes = (int *)contig_lens;
    gi.seq_ids = (char **)contig_ids;
    gi.seq = seq;
    lmo_gwriter_write(b->gw, &gi);
    uint64_t bg = ((uint64_t)batch << LMO_BITS_GENOME_IDX) | ((uint64_t)gidx & LMO_MASK_GENOME_IDX);
    if (bg_out) *bg_out = bg;
    {
        uint8_t buf[10];
        size_t l = strlen(id);
        if (l > 65535) l = 65535;
        buf[0] = (uint8_t)(l >> 8);
        buf[1] = (uint8_t)l;
        fwrite(buf, 1, 2, b->fmap);
        fwrite(id, 1, l, b->fmap);
        for (int i = 0; i < 8; i++) buf[i] = (uint8_t)(bg >> (56 - 8 * i));
        fwrite(buf, 1, 8, b->fmap);
    }
    uint64_t shift = bg << LMO_BITS_NONE_IDX;
    const uint64_t MASK_NONE_IDX = (((uint64_t)1) << LMO_BITS_NONE_IDX) - 1;

    /* masking */
    uint64_t *kmers = (uint64_t *)malloc(sizeof(uint64_t) * M);
    int *loc_off = NULL, *locs = NULL;
    if (lmo_lh_mask(b->lh, seq, len, nskip ? skip : NULL, nskip, 1, kmers, &loc_off, &locs) != 0) return -1;
    int nloc = 0;
    uint32_t *sorted = (uint32_t *)malloc(sizeof(uint32_t) * (loc_off[M] + 1));
    for (int i = 0; i < M; i++) {
        if (kmers[i] == 0 || lmo_low_complexity(kmers[i], k)) { /* lib-index-build.go:1037-1046 */
            kmers[i] = 0;
            continue;
        }
        for (int j = loc_off[i]; j < loc_off[i + 1]; j++) sorted[nloc++] = (uint32_t)locs[j];
    }
    qsort(sorted, nloc, sizeof(uint32_t), cmp_u32);

    /* extra k-mers: per mask */
    kvvec *extra = (kvvec *)calloc(M, sizeof(kvvec));
    /* desert filling, lib-index-build.go:1094-1407 */
    {
        uint32_t max_desert = (uint32_t)o->max_desert;
        int seed_dist = o->seed_dist, seed_pos_r = o->seed_dist / 2;
        sorted[nloc++] = (uint32_t)(len - k) << 1; /* pseudo position */
        uint32_t pre = 0;
        int wcap = 0;
        uint64_t *klist = NULL;
        int *l2m = NULL, *l2mrc = NULL;
        uint64_t *wh = (uint64_t *)malloc(sizeof(uint64_t) * M);
        int *wtouched = (int *)malloc(sizeof(int) * M);
        for (int i = 0; i < M; i++) wh[i] = ~(uint64_t)0;
        for (int il = 0; il < nloc; il++) {
            uint32_t pos = sorted[il] >> 1;
            uint32_t d = pos - pre;
            if (d < max_desert) {
                pre = pos;
                continue;
            }
            int start = (int)pre - 1000, pos_of_pre = 1000;
            if (start < 0) {
                pos_of_pre += start;
                start = 0;
            }
            int end = (int)pos + 1000 + k;
            if (end > len) end = len;
            int pos_of_cur = pos_of_pre + (int)d;
            int wlen = end - start;
            if (wlen > wcap) {
                wcap = wlen * 2;
                klist = (uint64_t *)realloc(klist, sizeof(uint64_t) * 2 * wcap);
                l2m = (int *)realloc(l2m, sizeof(int) * wcap);
                l2mrc = (int *)realloc(l2mrc, sizeof(int) * wcap);
            }
            lmo_kiter it;
            int nk = 0;
            if (lmo_kiter_init(&it, seq + start, wlen, k) == 0) {
                uint64_t a, c;
                while (lmo_kiter_next(&it, &a, &c)) {
                    klist[2 * nk] = a;
                    klist[2 * nk + 1] = c;
                    nk++;
                }
            }
            lmo_lh_window_l2m(b->lh, seq + start, wlen, wh, wtouched, l2m, l2mrc);
            int _j = pos_of_pre + seed_dist;
            for (;;) {
                if (_j >= pos_of_cur) break;
                int _start = _j + 1, _end = _j - seed_pos_r;
                int ok = 0, _im = -1;
                uint64_t kmer = 0, kmer_pos = 0;
                for (; _j > _end; _j--) {
                    if (_j < 0 || _j >= nk) continue;
                    if (in_intervals(iv, niv, start + _j)) continue;
                    kmer = klist[_j << 1];
                    if (valid_seed_kmer(kmer, k)) {
                        _im = l2m[_j];
                        if (_im >= 0) {
                            kmer_pos = (uint64_t)(start + _j) << 1;
                            ok = 1;
                            break;
                        }
                    }
                    kmer = klist[(_j << 1) + 1];
                    if (valid_seed_kmer(kmer, k)) {
                        _im = l2mrc[_j];
                        if (_im >= 0) {
                            kmer_pos = ((uint64_t)(start + _j) << 1) | 1;
                            ok = 1;
                            break;
                        }
                    }
                }
                if (ok) {
                    kv_push(&extra[_im], kmer, kmer_pos);
                    _j += seed_dist;
                    continue;
                }
                if (_start >= pos_of_cur) break;
                _end = _start + seed_pos_r;
                if (_end >= pos_of_cur) _end = pos_of_cur - 1;
                for (_j = _start; _j < _end; _j++) {
                    if (_j < 0 || _j >= nk) continue;
                    if (in_intervals(iv, niv, start + _j)) continue;
                    kmer = klist[_j << 1];
                    if (valid_seed_kmer(kmer, k)) {
                        _im = l2m[_j];
                        if (_im >= 0) {
                            kmer_pos = (uint64_t)(start + _j) << 1;
                            ok = 1;
                            break;
                        }
                    }
                    kmer = klist[(_j << 1) + 1];
                    if (valid_seed_kmer(kmer, k)) {
                        _im = l2mrc[_j];
                        if (_im >= 0) {
                            kmer_pos = ((uint64_t)(start + _j) << 1) | 1;
                            ok = 1;
                            break;
                        }
                    }
                }
                if (ok) {
                    kv_push(&extra[_im], kmer, kmer_pos);
                    _j += seed_dist;
                    continue;
                }
                _j += seed_dist;
            }
            pre = pos;
        }
        free(klist);
        free(l2m);
        free(l2mrc);
        free(wh);
        free(wtouched);
    }

    /* seed values: normal + extra, then reversed copies (lib-index-build.go:696-890) */
    for (int i = 0; i < M; i++) {
        if (kmers[i] != 0) {
            for (int j = loc_off[i]; j < loc_off[i + 1]; j++) {
                uint64_t value = shift | ((((uint64_t)(uint32_t)locs[j]) << 1) & MASK_NONE_IDX);
                kv_push(&b->data[i], kmers[i], value);
            }
        }
        for (int64_t j = 0; j < extra[i].n; j++) {
            uint64_t value = shift | ((extra[i].v[j].val << 1) & MASK_NONE_IDX);
            kv_push(&b->data[i], extra[i].v[j].kmer, value);
        }
    }
    for (int i = 0; i < M; i++) {
        if (kmers[i] != 0) {
            uint64_t rev = lmo_kmer_reverse(kmers[i], k);
            int minj = lmo_lh_mask_kmer_argmin(b->lh, rev);
            for (int j = loc_off[i]; j < loc_off[i + 1]; j++) {
                uint64_t value = shift | (((((uint64_t)(uint32_t)locs[j]) << 1) | 1) & MASK_NONE_IDX);
                kv_push(&b->data[minj], rev, value);
            }
        }
        for (int64_t j = 0; j < extra[i].n; j++) {
            uint64_t rev = lmo_kmer_reverse(extra[i].v[j].kmer, k);
            int minj = lmo_lh_mask_kmer_argmin(b->lh, rev);
            uint64_t value = shift | (((extra[i].v[j].val << 1) | 1) & MASK_NONE_IDX);
            kv_push(&b->data[minj], rev, value);
        }
        free(extra[i].v);
    }
    free(extra);
    free(sorted);
    free(kmers);
    free(loc_off);
    free(locs);
    free(skip);
    free(iv);
    free(seq);
    b->ngenomes++;
    b->input_bases += gsize;
    return 0;
}

static int cmp_kvpair(const void *a, const void *b) {
    const kvpair *x = (const kvpair *)a, *y = (const kvpair *)b;
    if (x->kmer != y->kmer) return x->kmer < y->kmer ? -1 : 1;
    return x->val < y->val ? -1 : x->val > y->val;
}

int lmo_builder_finish(lmo_builder *b) {
    const lmo_build_opt *o = &b->opt;
    int M = o->masks;
    if (b->gw) lmo_gwriter_close(b->gw);
    fclose(b->fmap);
    char p[4096];
    snprintf(p, sizeof p, "%s/genomes.chunks.bin", b->dir);
    FILE *f = fopen(p, "wb");
    if (f) { /* repeated: u64 n, n x u64 batch+index, big-endian (lib-index-build.go:1795-1806) */
        int at = 0;
        for (int l = 0; l < b->nchunk_lists; l++) {
            uint8_t buf[8];
            uint64_t n = (uint64_t)b->chunk_list_len[l];
            for (int i = 0; i < 8; i++) buf[i] = (uint8_t)(n >> (56 - 8 * i));
            fwrite(buf, 1, 8, f);
            for (int j = 0; j < b->chunk_list_len[l]; j++) {
                uint64_t v = b->chunk_keys[at++];
                for (int i = 0; i < 8; i++) buf[i] = (uint8_t)(v >> (56 - 8 * i));
                fwrite(buf, 1, 8, f);
            }
        }
        fclose(f);
    }
    free(b->chunk_keys);
    free(b->chunk_list_len);
    int nbatches = (b->ngenomes + o->batch_size - 1) / o->batch_size;
    int mask_prefix = b->lh->p;
    int anchor_prefix = 1;
    {
        int ap = 0;
        int64_t x = o->partitions;
        while (x >= 4) {
            x >>= 2;
            ap++;
        }
        anchor_prefix = ap < 1 ? 1 : ap; /* lib-index-search.go:469 */
    }
    int chunks = o->chunks;
    int chunk_size = (M + chunks - 1) / chunks;
    int nchunks_written = 0;
    for (int c = 0; c < chunks; c++) {
        int begin = c * chunk_size, end = begin + chunk_size;
        if (end > M) end = M;
        if (begin >= end) break;
        int nm = end - begin;
        lmo_kv_rec **recs = (lmo_kv_rec **)calloc(nm, sizeof(lmo_kv_rec *));
        int *nrecs = (int *)calloc(nm, sizeof(int));
        uint64_t **valbufs = (uint64_t **)calloc(nm, sizeof(uint64_t *));
        for (int i = 0; i < nm; i++) {
            kvvec *d = &b->data[begin + i];
            if (d->n == 0) continue;
            /* values of a k-mer are kept in a deterministic order (sorted); the reference's order is
             * map/merge dependent and never observable after anchor sorting */
            qsort(d->v, d->n, sizeof(kvpair), cmp_kvpair);
            int nr = 0;
            for (int64_t j = 0; j < d->n; j++)
                if (j == 0 || d->v[j].kmer != d->v[j - 1].kmer) nr++;
            recs[i] = (lmo_kv_rec *)malloc(sizeof(lmo_kv_rec) * nr);
            valbufs[i] = (uint64_t *)malloc(sizeof(uint64_t) * d->n);
            int r = -1;
            for (int64_t j = 0; j < d->n; j++) {
                valbufs[i][j] = d->v[j].val;
                if (j == 0 || d->v[j].kmer != d->v[j - 1].kmer) {
                    r++;
                    recs[i][r].kmer = d->v[j].kmer;
                    recs[i][r].vals = &valbufs[i][j];
                    recs[i][r].nvals = 0;
                }
                recs[i][r].nvals++;
            }
            nrecs[i] = nr;
        }
        snprintf(p, sizeof p, "%s/seeds/chunk_%03d.bin", b->dir, c);
        lmo_kv_write(p, o->k, begin, nm, recs, nrecs, mask_prefix, anchor_prefix, nbatches);
        nchunks_written++;
        for (int i = 0; i < nm; i++) {
            free(recs[i]);
            free(valbufs[i]);
            free(b->data[begin + i].v);
            b->data[begin + i].v = NULL;
        }
        free(recs);
        free(nrecs);
        free(valbufs);
    }
    snprintf(p, sizeof p, "%s/info.toml", b->dir);
    f = fopen(p, "w");
    if (f) {
        fprintf(f,
                "# Index format\nmain-version = 3\nminor-version = 5\n# LexicHash\nmax-K = %d\nmasks = %d\n"
                "rand-seed = %lld\n# Seed distance\nmax-seed-dist = %d\nseed-dist-in-desert = %d\n"
                "# Seeds (k-mer-value data) files\nchunks = %d\nindex-partitions = %d\n# Input genomes\n"
                "input-genomes = %d\n# Input bases\ninput-bases = %lld\n# Genome data.\ngenomes = %d\n"
                "genome-batch-size = %d\ngenome-batches = %d\ncontig-interval = %d\nsoft-masking = false\n"
                "max-kmer-freq = 0\n",
                o->k, o->masks, (long long)o->rand_seed, o->max_desert, o->seed_dist, nchunks_written, o->partitions,
                b->ninput, (long long)b->input_bases, b->ngenomes, o->batch_size, nbatches, o->contig_interval);
        fclose(f);
    }
    lmo_lh_free(b->lh);
    free(b->data);
    free(b->dir);
    free(b);
    return 0;
}
