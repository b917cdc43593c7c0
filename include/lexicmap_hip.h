/*
 * lexicmap_hip.h — C-ABI of the MI355X-native `lexicmap search` hot path (liblexicmap_hip.so).
 *
 * The reference (shenwei356/LexicMap) is pure Go built with CGO_ENABLED=0 and has no FFI of its own
 * (lexicmap/build.sh:7); this header is the seam a cgo shim in lexicmap/cmd would bind (INTEGRATION.md shows it).
 * Each entry point names the reference interface it replaces (paths relative to lexicmap/cmd/).
 *
 * Conventions
 *   - plain C types only; every function returns lm_status (0 = LM_OK) unless noted; lm_last_error() gives the text.
 *   - inputs are borrowed for the duration of the call (Go may move memory afterwards); outputs are callee-allocated
 *     and released with the matching *_free (mirrors the reference's Recycle* hand-back, lib-index-search.go:1170).
 *   - "no hit" is success with zero rows (the reference returns (nil,nil): lib-index-search.go:1669-1672).
 *   - the library needs a gfx950 GPU: every entry point that computes fails with LM_ERR_NO_DEVICE otherwise. There
 *     is no CPU fallback.
 */
#ifndef LEXICMAP_HIP_H
#define LEXICMAP_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int lm_status;
enum {
    LM_OK = 0,
    LM_ERR_IO = 1,        /* missing / broken index file (reference: checkError -> exit, util-cli.go:35) */
    LM_ERR_FORMAT = 2,    /* magic / version mismatch (kv-data.go:54-64, genome.go:58-71) */
    LM_ERR_OPTION = 3,    /* option out of range (search.go:159-229, lib-index-search.go:483-485) */
    LM_ERR_NO_DEVICE = 4, /* no HIP device / not gfx950 */
    LM_ERR_HIP = 5,       /* HIP runtime error */
    LM_ERR_NOMEM = 6,
    LM_ERR_ARG = 7
};

/* Search options: IndexSearchingOptions (lib-index-search.go:57-106) + SeqComparatorOptions as wired by
 * search.go:305-382.  Defaults = the flag defaults of search.go:631-731. */
typedef struct lm_options {
    int32_t min_prefix;          /* -p/--seed-min-prefix 15 */
    int32_t min_single_prefix;   /* -P/--seed-min-single-prefix 17 */
    int32_t top_n_genomes;       /* -n 0 */
    int32_t top_n_chains;        /* -N 0 */
    double max_gap;              /* --seed-max-gap 50 */
    double max_distance;         /* --seed-max-dist 1000 */
    int32_t ext_len;             /* --align-ext-len 1000 */
    int32_t ext_len2;            /* 50, hard-coded at search.go:325 */
    double min_qcov_per_genome;  /* -Q 0 */
    double max_evalue;           /* -e 10 */
    int32_t output_seq;          /* -a/--all */
    int32_t align_max_gap;       /* --align-max-gap 20 */
    int32_t align_band;          /* --align-band 100 */
    int32_t align_min_match_len; /* -l 50 */
    double align_min_pident;     /* -i 70 */
    double min_qcov_per_hsp;     /* -q 0 */
    /* sharding of the genome set across ranks: this process loads genomes with
     * (dense genome number % shard_count) == shard_rank. shard_count <= 1 loads everything. */
    int32_t shard_rank, shard_count;
    int64_t total_bases_override; /* >0: e-value database size shared by all shards (info.toml input-bases) */
} lm_options;

typedef struct lm_index lm_index;

typedef struct lm_index_info {
    int32_t k, masks, mask_prefix, anchor_prefix;
    int64_t total_bases;   /* info.toml input-bases: the e-value database size (lib-index-search.go:1918) */
    int64_t genomes;       /* genomes resident on this device */
    int64_t seeds;         /* (k-mer,value) pairs resident on this device */
    int64_t genome_bases;  /* concatenated bases resident on this device */
    int64_t hbm_bytes;     /* device memory held by the index image (seeds + genomes + tables) */
    int64_t seed_bytes;    /* device memory of the packed seed image alone (partition tables + key and value streams) */
    int64_t outlier_seeds; /* seeds kept in the flat 16-byte form (k-mer does not start with its mask's prefix) */
    int32_t key_bits, val_bits, partition_bases; /* packed seed layout: bits per k-mer remainder / value, bases per partition */
    int32_t pad;
} lm_index_info;

/* search.go:631-731 flag defaults */
void lm_options_default(lm_options *opt);

/* Replaces NewIndexSearcher(dir, opt) + SetSeqCompareOptions (lib-index-search.go:237-757, :217): reads info.toml,
 * masks.bin, seeds/chunk_*.bin(.idx), genomes/batch_NNNN/genomes.bin(.idx), genomes.map.bin and builds the HBM image on
 * HIP device `device`.  Limits of this build: at most 65535 masks (the reference's default is 20 000, 40 000 before
 * v0.6.0), genomes of at most 2^28 bases (the reference's own limit, lib-index-build.go:421-425). */
lm_status lm_index_open(const char *dir, const lm_options *opt, int device, lm_index **out);
/* Synthetic genome set + seed index generated directly in HBM (benchmark input; nothing in the reference corresponds to
 * it — index building is out of the hot-path scope).  Genome g belongs to family g % families; genomes >= families are
 * mutated copies (substitution rate U(0,max_div), indel shifts at a tenth of that) of the family ancestor.  Honors
 * opt->shard_rank/shard_count.  See lexicmap_amd/csrc/lm_builder.hip for what is exact and what is simplified. */
typedef struct lm_synth_spec {
    int32_t k;            /* 31 */
    int32_t masks;        /* 20000 (index.go:560) */
    int64_t mask_seed;
    int64_t genomes;      /* genomes in the whole set */
    int32_t genome_len;   /* bases per genome (one contig) */
    int32_t families;
    double max_div;
    int64_t seed;
    int32_t max_desert;   /* 100 (index.go:582) */
    int32_t seed_dist;    /* 50 (index.go:584) */
} lm_synth_spec;
lm_status lm_index_build_synthetic(const lm_synth_spec *spec, const lm_options *opt, int device, lm_index **out);
/* bases [start, start+len) of local genome `local_genome` as ASCII (used to derive synthetic queries) */
lm_status lm_index_fetch(lm_index *idx, int64_t local_genome, int64_t start, int64_t len, uint8_t *out);
/* Writes the resident (unsharded) index to `dir`: info.toml, seeds/chunk_NNN.bin (+ .idx, kv/kv-data.go:126-602) in at most
 * `chunks` files, genomes/batch_NNNN/genomes.bin (+ .idx, genome/genome.go:217-357), genomes.map.bin and genomes.chunks.bin in the reference's
 * on-disk format; masks.bin in THIS build's own layout (LMMASKS1: lexichash's file layout is not in the reference tree), so
 * lm_index_open and the oracle read the result back, the reference's Go binary does not.  info.toml carries the format
 * version, k, masks, chunk files, partitions, genome counts, input bases and the contig interval; the build-time settings it
 * does not know (rand-seed, the seed-distance settings, soft-masking, max-kmer-freq) are written as the reference's defaults
 * and are not read by a search.  Used to time the loader at benchmark scale on GPU-built sets.
 * genomes.chunks.bin (the chunk lists of genomes split at --max-genome, lib-index-build.go:1787-1808) is written too - empty
 * when no genome was split.  A shard (shard_count > 1) is refused. */
lm_status lm_index_save(lm_index *idx, const char *dir, int chunks);
/* Replaces (*Index).Close (lib-index-search.go:760) */
void lm_index_close(lm_index *idx);
lm_status lm_index_get_info(const lm_index *idx, lm_index_info *info);
/* masks of the index (lexichash.LexicHash.Masks), borrowed until close */
const uint64_t *lm_index_masks(const lm_index *idx);
/* (k-mer, value) pairs stored under one mask (normal seeds, then reversed seeds; each part ascending by k-mer), values
 * in the reference layout batch:17|genome:17|pos:28|rc:1|reversed:1: kv.Reader.ReadDataOfAMaskAsList
 * (kv/kv-reader.go:762), what `lexicmap utils kmers --mask` prints (kmers.go:101-180). Call with cap = 0 (or both arrays
 * NULL) for the count; 0 < cap < count returns LM_ERR_ARG with *n = count and writes nothing. */
lm_status lm_index_mask_seeds(lm_index *idx, int32_t mask, uint64_t *kmers, uint64_t *vals, size_t cap, size_t *n);
/* text of the last error on this handle, or of the last failed lm_index_open when idx == NULL */
const char *lm_last_error(const lm_index *idx);

/* ---------------------------------------------------------------------------------------------------------
 * Whole-path entry point: replaces the per-query goroutines calling (*Index).Search (search.go:548-608,
 * lib-index-search.go:1191-2940) with one batched call. */
typedef struct lm_query {
    const uint8_t *seq; /* upper-case bases (the host upper-cases, search.go:580-587) */
    uint32_t len;
} lm_query;

/* One HSP row = what the TSV printer consumes (search.go:468-523); coordinates 0-based inclusive. */
typedef struct lm_hsp {
    uint32_t query;        /* index into the batch */
    uint32_t hits;         /* number of subject genomes of this query ("hits" column) */
    uint64_t batch_genome; /* batch<<17 | genome index (key of genomes.map.bin) */
    double qcov_genome;    /* qcovGnm */
    int32_t cls, hsp;      /* 1-based counters as printed */
    int32_t seq_idx, nseqs, seq_len, nchunks, chunk_idx;
    int32_t rc;            /* subject strand '-' */
    double qcov_hsp;
    int32_t aligned_length;
    double pident;
    int32_t gaps;
    int32_t qbegin, qend, tbegin, tend;
    double evalue;
    int32_t bitscore, score, matched_bases;
    const char *genome_id, *seq_id;          /* borrowed from the index, valid until lm_index_close */
    const char *cigar, *qseq, *sseq, *align; /* only with output_seq; owned by the result batch */
} lm_hsp;

typedef struct lm_stage_stats { /* measured work per batch (H, A, C ...) */
    int64_t query_bases, query_kmers;
    int64_t seed_lookups;      /* (query,mask,direction) probes issued */
    int64_t seed_values;       /* H: seed values returned */
    int64_t anchors_raw;       /* anchors assembled (values x query locations) */
    int64_t genome_pairs;      /* (query,genome) pairs chained */
    int64_t anchors_cleared;   /* A: anchors after de-duplication */
    int64_t chains;            /* C: chains sent to alignment */
    int64_t window_bases;      /* sum of target window lengths */
    int64_t pa_anchors;        /* pseudo-alignment anchors */
    int64_t hsps_aligned;      /* WFA problems */
    int64_t wfa_retries;
    int64_t rows;              /* HSP rows emitted */
    int64_t aligned_bases;     /* sum of alenHSP over emitted rows */
    double ms_mask, ms_lookup, ms_chain, ms_window, ms_pseudo, ms_glue, ms_extend_wfa, ms_finalize, ms_total;
} lm_stage_stats;

typedef struct lm_result lm_result;

/* queries resident in HBM (so that a timed region can exclude the PCIe upload) */
typedef struct lm_qbatch lm_qbatch;
lm_status lm_qbatch_upload(lm_index *idx, const lm_query *queries, size_t nq, lm_qbatch **out);
void lm_qbatch_free(lm_qbatch *qb);
lm_status lm_search_resident(lm_index *idx, lm_qbatch *qb, lm_result **out);
/* = upload + search_resident */
lm_status lm_search_batch(lm_index *idx, const lm_query *queries, size_t nq, lm_result **out);
size_t lm_result_rows(const lm_result *res, const lm_hsp **rows); /* rows grouped by query, in output order */
void lm_result_stats(const lm_result *res, lm_stage_stats *stats);
void lm_result_free(lm_result *res); /* RecycleSearchResults, lib-index-search.go:1170 */
/* search.go:468-523: one TSV line (no newline); returns the length that was/would be written */
int lm_format_row(const lm_hsp *row, const char *query_id, uint32_t qlen, int more_columns, char *buf, size_t buflen);
/* the same line with the printer's two switches (search.go:483-520): LM_ROW_ALL = -a/--all (CIGAR, qseq, sseq, align
 * columns), LM_ROW_SSEQ_IDX = --show-sseq-idx (sseqid as c<chunk>/<chunks>:s<seq>/<seqs>:<id>, :483-494) */
#define LM_ROW_ALL 1
#define LM_ROW_SSEQ_IDX 2
int lm_format_row_ex(const lm_hsp *row, const char *query_id, uint32_t qlen, int flags, char *buf, size_t buflen);
/* The printer's loop over a batch (search.go:468-523; one writer goroutine in the reference): every row as lm_format_row_ex
 * writes it, a newline after each, in row order, formatted by the host threads into ONE buffer (*text, *len bytes, released
 * with lm_free).  query_ids / query_lens: the id and length of batch query i at [i], nq of them (rows[].query indexes them). */
lm_status lm_format_rows(const lm_hsp *rows, size_t n, const char *const *query_ids, const uint32_t *query_lens, size_t nq,
                         int flags, char **text, size_t *len);
/* the header line of the TSV (search.go:426-430), no newline; more_columns as in lm_format_row */
const char *lm_tsv_header(int more_columns);

typedef struct lm_stage lm_stage; /* host arrays of a stage-level call, released with lm_stage_free */

/* ---------------------------------------------------------------------------------------------------------
 * Genome-sharded indexes: every rank opens the index with its shard_rank / shard_count, searches the SAME
 * query batch, and the host gathers the per-rank rows (one all-gatherv of lm_hsp records).  lm_merge_sharded then
 * produces the reference's final order per query - genomes by the similarity (bitscore * pident) of their best HSP
 * cluster, descending (lib-index-search.go:2919-2921), each genome's rows as they were - and the global `hits`
 * (search.go:463,494): what `lexicmap utils merge-search-results` does for several indexes (merge-search-results.go:142-194).
 * rows[r] / nrows[r]: the rows of rank r, grouped by query in batch order (as lm_result_rows returns them).  Host-only;
 * idx (may be NULL) re-attaches genome_id / seq_id: every shard holds the names of all genomes.  cigar/qseq/sseq/align
 * are process-local and come back NULL.  Free with lm_result_free. */
lm_status lm_merge_sharded(lm_index *idx, const lm_hsp *const *rows, const size_t *nrows, int nshards, lm_result **out);

/* The ONE collective of the sharded search (north_star: "per-shard hit lists merged with a single RCCL all-gatherv over
 * xGMI"), behind the C-ABI so that the Go host needs nothing else: a gatherv of lm_hsp records to the merging rank - an
 * all-gather of the row counts (8 bytes per rank), then one group of point-to-point transfers into the root ((N-1) payloads
 * over the root's xGMI links; the ranks that do not merge receive nothing).  What the gathered rows are merged into:
 * lm_merge_sharded above (lib-index-search.go:2919-2921, merge-search-results.go:142-194).
 *   lm_comm_unique_id: rank 0 makes the 128-byte id (an ncclUniqueId) and hands it to the other ranks by the host's own
 *     means (the Go host: a file, a socket or its launcher's environment; bench.py: torch.distributed broadcast);
 *   lm_comm_init: every rank, once per process - one process per GPU, `device` = the GPU of this rank's index handle;
 *   lm_gather_rows: rows / n = this rank's rows (host memory, e.g. lm_result_rows).  nrows[lm_comm_size] receives every
 *     rank's count on every rank.  On `root`, *all_rows = the rows of rank 0, 1, ... back to back (pointer columns cleared:
 *     they are addresses of other processes; lm_merge_sharded re-attaches genome_id / seq_id), owned by the communicator
 *     and valid until its next call; elsewhere *all_rows = NULL.  All ranks must call it, in the same order.
 * RCCL is bound at run time (librccl.so.1; LM_RCCL_LIB overrides): a single-GPU user never loads it. */
#define LM_COMM_ID_BYTES 128
typedef struct lm_comm lm_comm;
lm_status lm_comm_unique_id(uint8_t id[LM_COMM_ID_BYTES]);
lm_status lm_comm_init(const uint8_t id[LM_COMM_ID_BYTES], int nranks, int rank, int device, lm_comm **out);
void lm_comm_free(lm_comm *comm);
int lm_comm_rank(const lm_comm *comm);
int lm_comm_size(const lm_comm *comm);
const char *lm_comm_last_error(const lm_comm *comm); /* comm == NULL: the last failed lm_comm_unique_id / lm_comm_init of this thread */
lm_status lm_gather_rows(lm_comm *comm, const lm_hsp *rows, size_t n, int root, const lm_hsp **all_rows, size_t *nrows);
/* lm_gather_rows + lm_merge_sharded in one call, the merge on the DEVICE: the other ranks' rows are received into device memory
 * in rank order, the root's own are uploaded beside them, the final order (the reference's, as lm_merge_sharded makes it) and the
 * global `hits` are computed there and downloaded once.  On `root`: *merged = `*total` rows in output order with genome_id /
 * seq_id re-attached from idx (the root's handle; NULL: left NULL), owned by the communicator and valid until its next call;
 * elsewhere *merged = NULL and *total = 0.  All ranks must call it, in the same order. */
lm_status lm_gather_merge_rows(lm_comm *comm, lm_index *idx, const lm_hsp *rows, size_t n, int root, const lm_hsp **merged,
                               size_t *total);
/* The merging rank's part of lm_gather_merge_rows by itself: d_rows = the rows of shard 0, 1, ... back to back in DEVICE memory
 * (nrows[r] each), merged on the device on the communicator's stream (a single-rank communicator will do), downloaded once,
 * names re-attached.  *merged / *total as above.  (How bench.py times the merge of N shards' rows on one GPU.) */
lm_status lm_merge_sharded_device(lm_comm *comm, lm_index *idx, const void *d_rows, const size_t *nrows, int nshards,
                                  const lm_hsp **merged, size_t *total);

/* -n/--top-n-genomes with a sharded index: the cut of lib-index-search.go:1781-1805 is over the genomes of ALL shards.
 *   1. every rank: lm_search_scores -> its candidates, per query at most top_n (query, genome, Chainer score)
 *   2. host: gather the candidates of all ranks; lm_topn_merge -> the global top-N per query (score descending, ties by
 *      genome key ascending: the total order this build uses where the reference's unstable sort leaves ties open)
 *   3. every rank: lm_search_resident_keep with that list instead of its local cut.
 * Unsharded indexes do the cut inside lm_search_resident. lm_topn_merge output arrays are released with lm_free. */
lm_status lm_search_scores(lm_index *idx, lm_qbatch *qb, lm_stage **out, size_t *n, const uint32_t **query,
                           const uint64_t **batch_genome, const float **score);
lm_status lm_topn_merge(int nshards, const uint32_t *const *query, const uint64_t *const *batch_genome,
                        const float *const *score, const size_t *n, int top_n, uint32_t **out_query,
                        uint64_t **out_batch_genome, size_t *out_n);
lm_status lm_search_resident_keep(lm_index *idx, lm_qbatch *qb, const uint32_t *keep_query, const uint64_t *keep_batch_genome,
                                  size_t nkeep, lm_result **out);
void lm_free(void *p);

/* Genome whitelist: the `genomeIds` argument of (*Index).Search (lib-index-search.go:1191,1396,1425-1489) - seeds of other
 * genomes are ignored when anchors are assembled.  The reference derives the same kind of per-genome keep flag from
 * taxids (-t/--taxids, LCA tests against taxdump: host-side, cached per genome): the host evaluates those and passes the
 * resulting genome keys here.  keys = batch<<17|index (genomes.map.bin); n = 0 clears the filter.  Applies to the calls
 * that follow on this handle. */
lm_status lm_index_set_genome_filter(lm_index *idx, const uint64_t *batch_genome_keys, size_t n);

/* ---------------------------------------------------------------------------------------------------------
 * Stage-level entry points (inner seams) used by the parity tests. All outputs are host arrays owned
 * by the returned lm_stage object. */
void lm_stage_free(lm_stage *s);

/* lexichash MaskKnownDistinctPrefixes + low-complexity zeroing (lib-index-search.go:1212-1238):
 * kmers[nq*M]; loc_off[nq*M+1] CSR into locs[] (pos<<1|strand ascending). */
lm_status lm_mask_batch(lm_index *idx, const lm_query *queries, size_t nq, lm_stage **out, const uint64_t **kmers,
                        const int64_t **loc_off, const int32_t **locs);

/* reverse re-bucketing + seed lookup + anchor assembly (lib-index-search.go:1268-1569) and, per (query,genome),
 * ClearSubstrPairs + Chainer.Chain (:1702-1775).
 * pairs: npairs records sorted by (query, batch_genome); raw anchors and cleared anchors as CSR. */
typedef struct lm_anchor {
    int32_t qbegin, tbegin;
    uint8_t len, trc, qrc, pad;
} lm_anchor;
typedef struct lm_pair {
    uint32_t query;
    uint64_t batch_genome;
    int64_t raw_off, raw_n;         /* into raw anchors (sorted in the ClearSubstrPairs order) */
    int64_t clr_off, clr_n;         /* into cleared anchors */
    float score;                    /* Chainer.Chain best score */
    int64_t chain_off, chain_n;     /* chains of this pair: chain_ptr[chain_off .. chain_off+chain_n] */
} lm_pair;
lm_status lm_seed_chain_batch(lm_index *idx, const lm_query *queries, size_t nq, lm_stage **out, size_t *npairs,
                              const lm_pair **pairs, const lm_anchor **raw, const lm_anchor **cleared,
                              const int64_t **chain_ptr, const int32_t **chain_idx);

/* SeqComparator.Index + Compare (lib-seq_compare.go:115-159,335-522) for explicit (query, target window) problems:
 * problem i compares queries[qidx[i]] over [qbegin[i], qend[i]] with targets[i]. Results CSR by problem. */
typedef struct lm_chain2 {
    int32_t qbegin, qend, tbegin, tend;
    int32_t nanchors, matched_bases, aligned_bases_q, aligned_bases_t;
    double pident;
} lm_chain2;
lm_status lm_pseudoalign_batch(lm_index *idx, const lm_query *queries, size_t nq, const lm_query *targets,
                               const uint32_t *qidx, const uint32_t *qbegin, const uint32_t *qend, size_t nproblems,
                               lm_stage **out, const int64_t **res_off, const lm_chain2 **res);

/* wfa.Aligner.Align(q,t) with DefaultPenalties, global, AdaptiveReduction(DefaultAdaptiveOption)
 * (lib-index-search.go:1910-1911,2261,2528). ops CSR: op<<32|n in forward order. */
typedef struct lm_wfa {
    int32_t status; /* 0 ok, 2 no match op */
    int32_t score;  /* WFA penalty score */
    int32_t qbegin, qend, tbegin, tend;
    uint32_t align_len, matches, gaps, gap_regions;
    int64_t ops_off;
    int32_t nops;
} lm_wfa;
lm_status lm_wfa_batch(lm_index *idx, const lm_query *q, const lm_query *t, size_t n, lm_stage **out,
                       const lm_wfa **res, const uint64_t **ops);

/* ---------------------------------------------------------------------------------------------------------
 * Measurement support: per-kernel HIP-event timing on the library's own stream. */
typedef struct lm_kernel_time {
    const char *name;
    int64_t launches;
    double total_ms;
    int64_t bytes;   /* algorithmic bytes accounted by the host for these launches */
} lm_kernel_time;
void lm_profile_enable(lm_index *idx, int on);
void lm_profile_reset(lm_index *idx);
/* exclusive != 0: the searches that follow run their kernels one after the other (no overlapped streams), so that the
 * per-kernel HIP-event times are exclusive times; results are unchanged. Measurement only. */
void lm_profile_exclusive(lm_index *idx, int exclusive);
size_t lm_profile_get(lm_index *idx, const lm_kernel_time **out);
/* Measurement only: waits for the device, launches the empty kernel lm::k_profile_mark(id) and waits again - a boundary a
 * rocprofv3 trace or counter pass can be cut at (tools/summarize_rocprof.py keeps the dispatches between the first two marks). */
void lm_profile_mark(lm_index *idx, int id);
/* Measurement only: re-reads the LM_* experiment switches (which a handle reads once, when it is opened or built) from the
 * environment, so that one resident index can be timed under several settings (bench.py --ab).  Results are unchanged by any
 * switch; the alignment scratch of the handle is dropped so that it is re-cut under the new settings. */
void lm_tuning_reload(lm_index *idx);

#ifdef __cplusplus
}
#endif
#endif
