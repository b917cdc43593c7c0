/*
 * lmo.h — CPU ORACLE for the `lexicmap search` hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference algorithm (shenwei356/LexicMap, Go).  It exists so that
 * the HIP product path (lexicmap_amd/csrc) can be checked bit-for-bit.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  Nothing under lexicmap_amd/ links, imports or calls it.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference/lexicmap/cmd).
 *
 * PARITY PINNING:
 *   - kv seed-file format + range-query semantics : pinned by kv/kv-data_test.go:31-361 (re-run in tests/)
 *   - 2-bit genome store + sub-sequence extraction : pinned by genome/genome_test.go:30-164
 *   - radix tree / range index / varint-GB        : pinned by their reference unit tests (restated)
 *   - e-value / bitscore arithmetic               : pinned by demo/q.gene.fasta.lexicmap.tsv rows
 *   - end-to-end HSP rows                          : near-golden against demo TSV goldens (own masks)
 *   - LexicHash masking (lexichash v0.5.5) and WFA (wfa v0.5.0) live in un-vendored Go modules that are NOT in
 *     /root/reference and cannot be executed here: for those two the oracle restates the published algorithms
 *     and is "PARITY UNPINNED" beyond the demo near-goldens (CIGAR rows of the top-2-genomes golden do pin
 *     the mismatch-only WFA paths).
 */
#ifndef LMO_H
#define LMO_H
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- value bit layout, lib-index-build.go:412-455 ---- */
#define LMO_BITS_GENOME_IDX 17
#define LMO_MASK_GENOME_IDX ((1u << 17) - 1)
#define LMO_BITS_NONE_IDX 30 /* 64-17-17 */
#define LMO_BITS_IDX 34
#define LMO_BITS_IDX_FLAGS 36

/* ---------- k-mer utilities (util/kmers.go, genome/genome.go:1427-1444) ---------- */
extern const uint8_t lmo_base2bit[256];
uint64_t lmo_kmer_encode(const uint8_t *s, int k);
void lmo_kmer_decode(uint64_t code, int k, char *out);
uint64_t lmo_kmer_revcomp(uint64_t code, int k);
uint64_t lmo_kmer_reverse(uint64_t code, int k); /* kmers.MustReverse: base-wise reversal */
int lmo_dust(uint64_t code, int k);              /* util/kmers.go:162-328, returns score>50 */
uint64_t lmo_ns(uint64_t b, int k);              /* util/kmers.go:434 */
int lmo_low_complexity(uint64_t kmer, int k);    /* ccc|ggg|ttt|dust — lib-index-search.go:1223-1238 */

/* k-mer iterator (lexichash/iterator semantics: every window, both strands) */
typedef struct {
    const uint8_t *s;
    int len, k;
    int idx; /* index of the window just returned */
    uint64_t fwd, rc, mask;
    int started;
} lmo_kiter;
int lmo_kiter_init(lmo_kiter *it, const uint8_t *s, int len, int k);
int lmo_kiter_next(lmo_kiter *it, uint64_t *kmer, uint64_t *kmer_rc);

/* ---------- LexicHash (lexichash v0.5.5 restated; call sites lib-index-search.go:430-478,1212-1220,1327-1336) --- */
typedef struct lmo_lh {
    int k, M, p;       /* p = mask prefix length = max(floor(log4 M),1) (lib-index-search.go:467) */
    uint64_t *masks;   /* sorted ascending */
    int *pfx_first;    /* [4^p+1] CSR: masks sharing each p-prefix (sorted masks => contiguous) */
} lmo_lh;
void lmo_gen_masks(int k, int M, int64_t seed, uint64_t *out);
lmo_lh *lmo_lh_new(int k, const uint64_t *masks, int M);
void lmo_lh_free(lmo_lh *lh);
int lmo_lh_write(const lmo_lh *lh, const char *file, int64_t seed);
lmo_lh *lmo_lh_read(const char *file, int64_t *seed);
/* Mask a sequence. kmers[M]; locs returned CSR (malloc'd, caller frees *loc_off and *locs).
 * skip: nskip inclusive [start,end] regions sorted by start. */
int lmo_lh_mask(const lmo_lh *lh, const uint8_t *seq, int len, const int *skip, int nskip, int check_shorter,
                uint64_t *kmers, int **loc_off, int **locs);
void lmo_lh_window_l2m(const lmo_lh *lh, const uint8_t *seq, int len, uint64_t *hashes, int *touched, int *l2m,
                       int *l2mrc);
/* argmin over masks sharing the p-prefix of kmer (lib-index-search.go:1328-1336) */
int lmo_lh_mask_kmer_argmin(const lmo_lh *lh, uint64_t kmer);

/* ---------- varint-GB (util/varint-GB.go) ---------- */
int lmo_put_uint64s(uint8_t *buf, uint64_t v1, uint64_t v2, uint8_t *ctrl);
int lmo_get_uint64s(uint8_t ctrl, const uint8_t *buf, int buflen, uint64_t *v1, uint64_t *v2);

/* ---------- kv seed files (kv/kv-data.go, kv-reader.go, kv-searcher2.go) ---------- */
typedef struct {
    uint64_t kmer;
    uint64_t *vals;
    int nvals;
} lmo_kv_rec; /* one distinct k-mer with its values; records of a mask sorted by kmer */
/* write one chunk file (+ .idx): recs[mask] arrays */
int lmo_kv_write(const char *file, int k, int mask_offset, int nmasks, lmo_kv_rec **recs, const int *nrecs,
                 int mask_prefix, int anchor_prefix, int nbatches);
typedef struct {
    int k, chunk_index, chunk_size, mask_prefix, anchor_prefix, use7;
    uint64_t **kv; /* per mask: interleaved kmer,value,... (kv-reader.go:762) */
    int64_t *kvlen; /* number of uint64 in kv[i] */
    int64_t **index; /* per mask: [4^anchor_prefix] first offset or -1 */
} lmo_kv_mem;
lmo_kv_mem *lmo_kv_load(const char *file);
void lmo_kv_free(lmo_kv_mem *m);
/* result of a search: flattened */
typedef struct {
    int iquery, iquery2;
    uint8_t len, is_suffix;
    int64_t val_off;
    int nvals;
} lmo_kv_sr;
typedef struct {
    lmo_kv_sr *sr;
    int n, cap;
    uint64_t *vals;
    int64_t nv, capv;
} lmo_kv_results;
void lmo_kv_results_free(lmo_kv_results *r);
/* kv-searcher2.go:105-323 */
int lmo_kv_search(const lmo_kv_mem *m, const uint64_t *kmers, int p, int check_flag, int reversed, lmo_kv_results *out);
/* kv-searcher2.go:326-549; kmersR CSR per mask of the chunk */
int lmo_kv_search2(const lmo_kv_mem *m, const uint64_t *kmersR, const int *kr_off, int p, int check_flag, int reversed,
                   lmo_kv_results *out);
/* .idx reader (kv-data.go:619-769): returns dense table per mask [2+2*4^a] or NULL for empty masks */
int lmo_kv_read_index(const char *file, int *k, int *chunk_index, int *chunk_size, int *mask_prefix, int *anchor_prefix,
                      uint64_t ***tables);

/* ---------- genome store (genome/genome.go) ---------- */
int lmo_seq2twobit(const uint8_t *s, int n, uint8_t *out); /* returns nbytes */
void lmo_twobit2seq(const uint8_t *b2, int bases, uint8_t *out);
typedef struct {
    char *id;
    int genome_size, len, nseqs;
    int *seq_sizes;
    char **seq_ids;
    const uint8_t *seq; /* concatenated ASCII (len bases) */
} lmo_genome_in;
typedef struct lmo_gwriter lmo_gwriter;
lmo_gwriter *lmo_gwriter_open(const char *file, uint32_t batch);
int lmo_gwriter_write(lmo_gwriter *w, const lmo_genome_in *g);
int lmo_gwriter_close(lmo_gwriter *w);
typedef struct {
    FILE *fh;
    uint32_t batch, n;
    uint64_t *offsets;
    uint32_t *bases;
} lmo_greader;
typedef struct {
    int genome_size, len, nseqs;
    int *seq_sizes;
    char **seq_ids;
    int64_t seq_offset; /* file offset of the (nbytes,bases) header of packed data */
    uint8_t *seq;       /* last extracted subsequence */
    int seqlen, seqcap;
} lmo_genome;
lmo_greader *lmo_greader_open(const char *file);
void lmo_greader_close(lmo_greader *r);
/* genome.go:931-1143; g==NULL on first call for a genome */
lmo_genome *lmo_subseq3(lmo_greader *r, int idx, int start, int end, lmo_genome *g);
void lmo_genome_free(lmo_genome *g);

/* ---------- anchors / chaining ---------- */
typedef struct {
    int32_t qbegin, tbegin;
    uint8_t len, trc, qrc, _pad;
} lmo_sub; /* SubstrPair, lib-index-search.go:805-817 */

int lmo_clear_subs(lmo_sub *subs, int n, int k); /* ClearSubstrPairs :864-990; returns new n */
void lmo_sort_subs(lmo_sub *subs, int n);        /* the sort used by ClearSubstrPairs, deterministic total order */

typedef struct {
    float max_gap, min_score, max_distance;
    int top_chains;
    float gap_lut[1]; /* unused placeholder */
} lmo_chain_opt;
/* Chainer.Chain lib-chaining.go:122-633. chains returned CSR (malloc'd). returns best score */
float lmo_chainer(const lmo_sub *subs, int n, float max_gap, float min_score, float max_distance, int top_chains,
                int **chain_off, int **chain_idx, int *nchains);
float lmo_seed_weight(float l);
float lmo_gap_score(float g);
double lmo_go_log2(double x);

typedef struct {
    int nanchors;
    double aligned_fraction;
    int matched_bases, aligned_bases_q, aligned_bases_t;
    double pident;
    int aligned_length, gaps;
    int qbegin, qend, tbegin, tend;
    int max_ext_len, tpos_offset_begin;
    int score, bitscore;
    double evalue;
    /* -a output */
    char *cigar, *qseq, *tseq, *align;
    int alive;
} lmo_chain2; /* Chain2Result lib-chaining2.go:106-135 */
typedef struct {
    int max_gap, min_score, min_align_len;
    double min_identity;
    int band_count, band_base;
    double heuristic_pident;
} lmo_chain2_opt;
/* Chainer2.Chain lib-chaining2.go:152-358; returns number of chains, *out malloc'd; totals optional */
int lmo_chainer2(const lmo_sub *subs, int n, const lmo_chain2_opt *opt, lmo_chain2 **out, int *aligned_q);
/* Chainer3.Chain lib-chaining3.go:111-299: returns 1 and fills qend/tend if a chain found */
int lmo_chainer3(const lmo_sub *subs, int n, int *qend, int *tend);
int lmo_trim_subs(lmo_sub *subs, int n, int k, float min_dist, int *start_out); /* TrimSubStrPairs; returns new n */

/* ---------- radix tree (tree/tree.go) ---------- */
typedef struct lmo_tree lmo_tree;
typedef struct {
    uint64_t key;
    uint32_t val;
} lmo_tree_entry;
lmo_tree *lmo_tree_new(int k);
void lmo_tree_free(lmo_tree *t);
void lmo_tree_insert(lmo_tree *t, uint64_t key, uint32_t v);
void lmo_tree_insert_batch(lmo_tree *t, lmo_tree_entry *e, int n); /* sorts e by key (stable) */
typedef struct {
    uint64_t kmer;
    uint8_t len_prefix;
    const uint32_t *vals;
    int nvals;
} lmo_tree_sr;
/* tree.go:441-527; results appended into *out (realloc'd), returns count */
int lmo_tree_search(const lmo_tree *t, uint64_t key, int p, lmo_tree_sr **out, int *cap);

/* ---------- pseudo-alignment (lib-seq_compare.go) ---------- */
typedef struct {
    int k, min_prefix;
    lmo_chain2_opt c2;
    double min_aligned_fraction, min_identity;
} lmo_cmp_opt;
typedef struct {
    lmo_cmp_opt opt;
    lmo_tree *tree;
} lmo_cmp;
lmo_cmp *lmo_cmp_new(const lmo_cmp_opt *opt);
void lmo_cmp_free(lmo_cmp *c);
int lmo_cmp_index(lmo_cmp *c, const uint8_t *s, int len); /* :115-159 */
/* :335-522; returns #chains (sorted by qbegin), 0 if none. also returns raw anchors post clear/trim if subs_out!=NULL */
int lmo_cmp_compare(lmo_cmp *c, uint32_t begin, uint32_t end, const uint8_t *t, int tlen, int query_len,
                    lmo_chain2 **chains, lmo_sub **subs_out, int *nsubs_out);
int lmo_coverage_len(int (*regions)[2], int n); /* :270-308 */

/* ---------- extension (lib-index-search-util.go:34-201) ---------- */
void lmo_extend_match(const uint8_t *seq1, int len1, const uint8_t *seq2, int len2, int start1, int end1, int start2,
                      int end2, int ext_len, int tbegin, int max_ext_len, int rc, int *o_start1, int *o_end1,
                      int *o_start2, int *o_end2, int *s1, int *e1, int *s2, int *e2);

/* ---------- WFA (wfa v0.5.0 restated: gap-affine x=4,o=6,e=2, global, WFA2 tie-breaking, wf-adaptive 10/50) -- */
typedef struct {
    uint64_t *ops; /* op<<32 | n */
    int nops;
    int qbegin, qend, tbegin, tend; /* 1-based, M-trimmed region */
    uint32_t align_len, matches, gaps, gap_regions;
    int score; /* wfa penalty score */
} lmo_wfa_result;
int lmo_wfa_align(const uint8_t *q, int qlen, const uint8_t *t, int tlen, int adaptive, lmo_wfa_result *res);
void lmo_wfa_result_free(lmo_wfa_result *r);
/* scoreAndEvalue lib-index-search-util.go:260-304 */
void lmo_score_evalue(const lmo_wfa_result *r, int qlen, int64_t total_bases, int *score, int *bitscore, double *evalue);

/* ---------- index build (synthetic-index writer; follows lib-index-build.go) ---------- */
typedef struct {
    int k, masks;
    int64_t rand_seed;
    int max_desert, seed_dist, chunks, partitions, batch_size, contig_interval;
    int max_genome; /* -g/--max-genome 20,000,000 (index.go:538): longer concatenations are split into genome chunks */
} lmo_build_opt;
void lmo_build_opt_default(lmo_build_opt *o);
typedef struct lmo_builder lmo_builder;
lmo_builder *lmo_builder_new(const char *outdir, const lmo_build_opt *opt);
int lmo_builder_set_masks(lmo_builder *b, const uint64_t *masks, int n); /* test hook: caller-provided mask set */
/* add one genome: contigs concatenated by the builder with contig_interval 'A's; when the concatenation would exceed
 * max_genome the genome is split into chunks, each stored as its own genome record with the same id and listed together
 * in genomes.chunks.bin (lib-index-build.go:1581-1658,1786-1808). Returns -2 when one contig alone exceeds max_genome
 * (the reference skips such genomes). */
int lmo_builder_add(lmo_builder *b, const char *id, int ncontigs, const char **contig_ids, const uint8_t **contigs,
                    const int *contig_lens);
int lmo_builder_finish(lmo_builder *b);

/* ---------- search (lib-index-search.go) ---------- */
typedef struct {
    int min_prefix, min_single_prefix;
    int top_n, top_n_chains;
    double max_gap, max_distance;
    int ext_len, ext_len2;
    double min_qcov_genome, max_evalue;
    int output_seq;
    /* SeqComparatorOptions / Chaining2Options, search.go:360-382 */
    int align_max_gap, align_band, align_min_match_len;
    double align_min_pident, min_qcov_hsp;
} lmo_search_opt;
void lmo_search_opt_default(lmo_search_opt *o);

typedef struct lmo_index lmo_index;
lmo_index *lmo_index_open(const char *dir, const lmo_search_opt *opt);
void lmo_index_close(lmo_index *idx);
int lmo_index_k(const lmo_index *idx);
int lmo_index_nmasks(const lmo_index *idx);
const uint64_t *lmo_index_masks(const lmo_index *idx);
int64_t lmo_index_total_bases(const lmo_index *idx);
const lmo_lh *lmo_index_lh(const lmo_index *idx);

/* One HSP row as consumed by the TSV printer (search.go:468-523) */
typedef struct {
    uint64_t batch_genome; /* batch<<17|genome */
    double qcov_genome;
    int cls, hsp;          /* 1-based counters as printed */
    int seq_idx, nseqs, seq_len, nchunks, chunk_idx;
    int rc;
    double qcov_hsp;
    int aligned_length;
    double pident;
    int gaps;
    int qbegin, qend, tbegin, tend; /* 0-based inclusive */
    double evalue;
    int bitscore, score;
    int matched_bases;
    char *cigar, *qseq, *tseq, *align; /* output_seq only */
    const char *genome_id, *seq_id;
} lmo_hsp;
typedef struct {
    lmo_hsp *rows;
    int n, cap;
    int ngenomes; /* "hits" */
    /* stage statistics */
    int64_t n_seed_values, n_anchors, n_genomes_seeded, n_chains;
} lmo_result;
int lmo_search(lmo_index *idx, const uint8_t *seq, int len, lmo_result *res);
void lmo_result_free(lmo_result *r);
/* search.go:468-523 row formatting */
int lmo_format_row(const lmo_hsp *h, const char *query_id, int qlen, int hits, int more_columns, char *buf, size_t buflen);

/* ---- stage-level entry points used by parity tests (flat arrays) ---- */
/* stage 1+2: mask + low-complexity zeroing. kmers[M], locs CSR */
int lmo_stage_mask(lmo_index *idx, const uint8_t *seq, int len, uint64_t *kmers, int **loc_off, int **locs);
/* stage 3..5: seed lookup + anchor assembly. Returns anchors sorted by (genome, clear-order). */
typedef struct {
    uint64_t genome;
    lmo_sub sub;
} lmo_anchor;
int64_t lmo_stage_anchors(lmo_index *idx, const uint64_t *kmers, const int *loc_off, const int *locs, lmo_anchor **out);

#ifdef __cplusplus
}
#endif
#endif
