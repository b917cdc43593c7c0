// lm_kernels.h — device-side record types and kernel launchers (see lm_kernels.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lm_algos.h"

namespace lm {

// HBM image of the index (DESIGN.md §HBM layout)
struct DevIndexView {
    int K, M, mask_prefix;
    const uint64_t *masks;      // [M] sorted
    const int32_t *pfx_first;   // [4^p + 1] masks sharing each p-base prefix
    // packed seed image: list md = 2*mask + direction (0 = prefix seeds, 1 = reversed/suffix seeds)
    int part_bases;             // a: bases of the partition index (the index's anchor prefix, kv-data.go:90-125)
    int P1;                     // 4^a + 1: stride of part_tab
    int key_bits;               // 2 (K - p - a)
    int gid_bits, pos_bits;     // value = local genome : gid_bits | position : pos_bits | strand : 1
    const uint64_t *pk_keys;    // bit stream, key_bits per seed
    const uint64_t *pk_vals;    // bit stream, gid_bits + pos_bits + 1 per seed
    const uint32_t *part_tab;   // [2M][P1] first seed of each partition, relative to md_off[md]; [P1-1] = list length
    const int64_t *md_off;      // [2M+1] first seed of each list
    const uint64_t *g_bg;       // [G] batch:17|genome:17 key of each local genome (genomes.map.bin key)
    const uint32_t *g_keep;     // genome whitelist: bit per local genome, or null (lib-index-search.go:1425-1489)
    // seeds whose k-mer does not start with its mask's p-base prefix (captures of genomes that lack the prefix:
    // tiny genomes only) keep the reference's flat form, per list sorted by (k-mer, value)
    const uint64_t *out_kmers;  // [No]
    const uint64_t *out_vals;   // [No] batch:17|genome:17|pos:28|rc:1|reversed:1 (lib-index-build.go:412-455)
    const int64_t *out_off;     // [2M+1]
    const uint8_t *gbits;       // 2-bit genomes, first base in bits 7-6 (genome.go:1480)
    const int64_t *g_off;       // [G] byte offset
    const int32_t *g_len;       // [G] concatenated length in bases
    const int64_t *batch_first; // [nbatches+1]
    int nbatches;
    int64_t ngenomes;           // local
    int shard_rank, shard_count;
    const int32_t *g2local;     // dense genome number -> local number (-1: other shard), or null: g % count == rank, g / count
};

struct Task { // one lexichash chain of one (query, genome): the pseudo-alignment problem
    uint32_t seg, q;
    int32_t g, rc;
    int32_t tBegin, tEnd, qBegin, qEnd;
    int32_t nseeds, wlen;
    int64_t woff;
    uint64_t bg; // batch:17|genome:17 key of the genome
};

struct HspIn { // extendMatch input (lib-index-search.go:2255,2522)
    uint32_t q;
    int32_t rc;
    int64_t woff;
    int32_t len1, len2;
    int32_t start1, end1, start2, end2;
    int32_t ext_len, tbegin, max_ext_len, pad;
};
struct HspExt {
    int32_t qs, qe, ts, te, s1, e1, s2, e2;
};

struct WfaIn {
    const uint8_t *q, *t;
    int32_t qlen, tlen;
    int64_t hdr_off, arena_off, arena_cap, ops_off;
    int32_t max_score, ops_cap;
};
struct WfaOut {
    LmWfaOut r;
    int32_t blast_score;
};

void launch_extract_kmers(hipStream_t st, const uint8_t *qseq, const int64_t *qoff, const int64_t *posoff, int nq, int K,
                          int64_t total_pos, uint64_t *keys_all, uint32_t *vals_all, uint64_t *keys_cmp,
                          uint32_t *vals_cmp, int32_t *nvalid);
void launch_fill_u32(hipStream_t st, uint32_t *p, int64_t n, uint32_t v);
void launch_mask(hipStream_t st, const uint64_t *keys_all, const int64_t *posoff, int nq, int M, int K,
                 const uint64_t *masks, uint64_t *out_kmers, int64_t *out_lo, int64_t *out_hi, uint32_t *first_mask);
void launch_lookup_prep(hipStream_t st, DevIndexView ix, const uint64_t *kmers, const int64_t *klo, const uint32_t *first_mask,
                        int64_t nqm, uint32_t *keys, uint32_t *slots, unsigned long long *counter);
void launch_lookup_count(hipStream_t st, DevIndexView ix, const uint64_t *kmers, const int64_t *klo, const int64_t *khi,
                         const uint32_t *skeys, const uint32_t *sslots, int64_t nlk, int min_prefix, uint32_t *counts,
                         int64_t *starts, int32_t *nscan, unsigned long long *stat_values, uint64_t *lkey, uint64_t *lrec);
// anchors with the lanes over the output (full-line stores, seeds read front to back); not under a genome whitelist
void launch_lookup_emit_flat(hipStream_t st, DevIndexView ix, const uint32_t *vals_all, const uint32_t *skeys, const uint32_t *sslots,
                             int64_t nlk, const uint32_t *counts, const int64_t *offs, const int64_t *starts, const uint64_t *lkey,
                             const uint64_t *lrec, uint64_t *outA, uint64_t *outB);
void launch_lookup_emit(hipStream_t st, DevIndexView ix, const uint64_t *kmers, const int64_t *klo, const int64_t *khi,
                        const uint32_t *vals_all, const uint32_t *skeys, const uint32_t *sslots, int64_t nlk,
                        const uint32_t *counts, const int64_t *offs, const int64_t *starts, const int32_t *nscan,
                        uint64_t *outA, uint64_t *outB);
void launch_chain1(hipStream_t st, const uint64_t *B, const int64_t *seg_off, int nseg, LmChainOpt opt, int K, LmSub *subs,
                   uint8_t *marks, uint64_t *msi, uint64_t *s2i, int8_t *dirs, uint8_t *visited, int32_t *chain_off_pool,
                   int32_t *chain_idx_pool, int32_t *seg_n, float *seg_score, int32_t *seg_nch, int32_t *big_list,
                   unsigned int *big_count);
void launch_task_count(hipStream_t st, const float *seg_score, const int32_t *seg_nch, const uint8_t *keep, int nseg,
                       float min_score, int32_t *ntask);
void launch_make_tasks(hipStream_t st, DevIndexView ix, const uint64_t *segA, const int64_t *seg_off, int nseg,
                       const LmSub *subs, const int32_t *chain_off_pool, const int32_t *chain_idx_pool,
                       const int32_t *ntask, const int64_t *task_off, const int64_t *qoff, int ext_len,
                       int32_t *order_scratch, Task *tasks);
void launch_task_wlen(hipStream_t st, const Task *tasks, int64_t ntasks, int32_t *wlen);
void launch_task_set_woff(hipStream_t st, Task *tasks, int64_t ntasks, const int64_t *woff);
void launch_extract_windows(hipStream_t st, DevIndexView ix, const Task *tasks, int64_t ntasks, const int32_t *only,
                            uint8_t *wbuf);
void launch_extract_windows_at(hipStream_t st, DevIndexView ix, const Task *tasks, const int32_t *idx, const int64_t *dest,
                               int64_t n, uint8_t *wbuf);
#define LM_TAB_BITS_MIN 12 /* bucket table over the leading bits of a query's sorted k-mers: 2^12 .. 2^20 buckets */
#define LM_TAB_BITS_MAX 20
void launch_build_cmp_tab(hipStream_t st, const uint64_t *keys_cmp, const int64_t *posoff, const int32_t *nvalid, int nq,
                          int K, const int64_t *tab_off, const int32_t *tab_bits, int64_t tab_words, uint32_t *tab);
void launch_sum_i32(hipStream_t st, const int32_t *v, int64_t n, unsigned long long *out);
// hashed 11-base prefix bitmap per query: 2^bits_log[q] bits at word bits_off[q] (sized by the host, ~16 bits per k-mer)
void launch_build_cmp_bits(hipStream_t st, const uint64_t *keys_cmp, const int64_t *posoff, const int32_t *nvalid, int nq,
                           int K, const int64_t *bits_off, const int32_t *bits_log, uint32_t *bits);
#define LM_PA_MAX_SEGS 8192 /* segments (each with its own counter) of the candidate list of k_pa_filter */
#define LM_PA_RANGE_SEGS 16 /* segments per range of task groups (= wavefronts of a k_pa_filter workgroup) */
#define LM_PA_GROUP 64 /* chain windows per k_pa_filter workgroup pass */
void launch_pa_filter(hipStream_t st, DevIndexView ix, const Task *tasks, int64_t ntasks, const uint8_t *wbuf,
                      const int64_t *posoff, const int32_t *nvalid, const uint32_t *cmp_bits, const int64_t *bits_off,
                      const int32_t *bits_log, int K, int min_prefix, unsigned long long *seg_count, int nseg, int64_t seg_cap,
                      uint64_t *cand, unsigned long long *group_counter, int ncu, int seg_by_group, bool roll = true);
void launch_pa_search(hipStream_t st, DevIndexView ix, const Task *tasks, const uint8_t *wbuf, const uint64_t *keys_cmp,
                      const uint32_t *vals_cmp, const int64_t *posoff, const int32_t *nvalid, const uint32_t *cmp_tab,
                      const int64_t *tab_off, const int32_t *tab_bits, int K, int min_prefix,
                      const unsigned long long *seg_count, int nseg, int64_t seg_cap, const uint64_t *cand,
                      unsigned long long *count, int64_t cap, uint64_t *outA, uint64_t *outB, int qbits, int tbits,
                      int xcd_map);
void launch_pa_task_off_sorted(hipStream_t st, const uint64_t *sortedA, int shift, int64_t total, int64_t ntasks,
                               int64_t *pa_off);
void launch_pa_chain(hipStream_t st, const uint64_t *B, const int64_t *pa_off, int64_t ntasks, int K, LmChain2Opt opt,
                     LmSub *subs, uint8_t *marks, uint64_t *msi, int32_t *stack, LmChain2 *out, int32_t *out_n,
                     int32_t *clr_n, int qbits, int tbits);
void launch_gather_chain2(hipStream_t st, const LmChain2 *in, const int64_t *pa_off, const int32_t *out_n,
                          const int64_t *res_off, int64_t ntasks, LmChain2 *out);
void launch_extend_count(hipStream_t st, const HspIn *hsps, int64_t n, const uint8_t *qseq, const int64_t *qoff,
                         const uint8_t *wbuf, int32_t *cap);
void launch_extend_wave_cap(hipStream_t st, const int32_t *cap, int64_t n, int32_t *wcap, int64_t nw);
int extend_grid_blocks(int64_t n);
void launch_extend(hipStream_t st, const HspIn *hsps, int64_t n, const uint8_t *qseq, const int64_t *qoff,
                   const uint8_t *wbuf, const int32_t *cap, const int64_t *woff, uint16_t *subs, int32_t *msi,
                   void *rows_pool, uint32_t *rstart_pool, HspExt *out);
// k_wfa_lean<nc>: persistent wavefronts with private scratch, <= 64 nc - 2 diagonals (status 3 beyond); nc = 2, 4, 8 or 16
int wfa_resident_blocks(int device, int seq_words, int nc, bool win, bool r16 = false);
// 16-bit ring cells (half the LDS per wavefront): whole-sequence kernels of 128 / 256 diagonals, sequences <= 12 000 bases
bool wfa_r16_ok(int seq_words, int nc, bool win);
void launch_wfa(hipStream_t st, const WfaIn *in, int64_t n, const int32_t *todo, int64_t ntodo, int nblocks,
                int32_t *hdr_pool, int64_t hdr_stride, uint8_t *arena_pool, int64_t arena_stride, uint64_t *ops_pool,
                unsigned int *queue, int seq_words, int want_ops, WfaOut *out, int nc, bool win, bool r16 = false,
                unsigned long long *dbg = nullptr); // (dbg: unused)

// wavefronts wider than the LDS ring (status 3 from launch_wfa): same algorithm with the ring in global memory
void launch_wfa_wide(hipStream_t st, const WfaIn *in, int64_t n, const int32_t *todo, int64_t ntodo, int32_t *hdr_pool,
                     int32_t *arena_pool, uint64_t *ops_pool, WfaOut *out);

} // namespace lm
