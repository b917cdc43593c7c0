/* tests/cabi_shim.c — `lexicmap search` as the Go host of INTEGRATION.md would run it, in plain C99 over the C-ABI.
 *
 * A Go toolchain is not available here, so the Go file of INTEGRATION.md cannot be compiled; this is the same program in C:
 * the flag set and checks of search.go:159-229 mapped onto lm_options, the reader loop of
 * search.go:548-608 as a BATCH loop (records upper-cased, records shorter than k counted and skipped, a flush to
 * lm_search_batch every --batch-bases bases / --batch-queries records, any number of input files, FASTA or FASTQ), and the
 * printer of search.go:426-533 (header line, -a/--all columns, --show-sseq-idx, the final "queries matched" log).
 * cgo relies on exactly what this file relies on: include/lexicmap_hip.h is valid C (cgo compiles the preamble with a C
 * compiler), every call links against liblexicmap_hip.so with C linkage.  tests/test_cabi_and_host.py compiles and links it
 * on the CPU box (-Wall -Wextra -pedantic -Werror; no run without a GPU); tests/test_gpu_shim.py runs it on the GPU
 * against the reference's own golden TSVs.
 *
 *   usage: cabi_shim -d <index dir> [flags] <query file> [<query file> ...]        (or: cabi_shim <index dir> <query file>)
 */
#define _POSIX_C_SOURCE 200809L
#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "lexicmap_hip.h"

typedef struct {
    char *id;
    char *seq;
    size_t len, cap;
} record;

typedef struct {
    record *r;
    size_t n, cap;
    size_t bases;
} batch;

static void die(const char *msg) {
    fprintf(stderr, "[ERRO] %s\n", msg); /* checkError (util-cli.go:35): message, exit code -1 */
    exit(255);
}

static void batch_push(batch *b, const char *id, const char *seq, size_t len) {
    if (b->n == b->cap) {
        b->cap = b->cap ? 2 * b->cap : 64;
        b->r = (record *)realloc(b->r, b->cap * sizeof *b->r);
        if (!b->r) die("out of memory");
    }
    record *r = &b->r[b->n++];
    r->id = strdup(id);
    r->seq = (char *)malloc(len + 1);
    if (!r->id || !r->seq) die("out of memory");
    for (size_t i = 0; i < len; i++) { /* search.go:580-587: lower case -> upper case, nothing else */
        char c = seq[i];
        r->seq[i] = (c >= 'a' && c <= 'z') ? (char)(c - ('a' - 'A')) : c;
    }
    r->seq[len] = 0;
    r->len = len;
    b->bases += len;
}

typedef struct {
    lm_index *h;
    FILE *out;
    int row_flags;
    unsigned long long total, matched, rows, bases, text_bytes;
    int flushes;
    double t_search, t_format, t_write;
} ctx;

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* SearchBatch + printResult for every record of the batch, in input order (search.go:437-533): the rows of the batch are
 * formatted by ONE call (lm_format_rows: the host threads, one buffer) and written with one fwrite */
static void flush(ctx *c, batch *b) {
    if (b->n == 0) return;
    lm_query *qs = (lm_query *)malloc(b->n * sizeof *qs);
    const char **ids = (const char **)malloc(b->n * sizeof *ids);
    uint32_t *lens = (uint32_t *)malloc(b->n * sizeof *lens);
    if (!qs || !ids || !lens) die("out of memory");
    for (size_t i = 0; i < b->n; i++) {
        qs[i].seq = (const uint8_t *)b->r[i].seq;
        qs[i].len = (uint32_t)b->r[i].len;
        ids[i] = b->r[i].id;
        lens[i] = (uint32_t)b->r[i].len;
    }
    lm_result *res = NULL;
    double t0 = now_s();
    if (lm_search_batch(c->h, qs, b->n, &res) != LM_OK) die(lm_last_error(c->h));
    double t1 = now_s();
    const lm_hsp *rows = NULL;
    const size_t m = lm_result_rows(res, &rows); /* grouped by query in batch order; within a query in final order */
    size_t j = 0;
    for (size_t i = 0; i < b->n; i++) { /* search.go:439-446: a query without rows is counted and prints nothing */
        c->total++;
        size_t k = j;
        while (k < m && rows[k].query == (uint32_t)i) k++;
        if (k > j) c->matched++;
        j = k;
    }
    if (j != m) die("rows of a query number outside the batch");
    char *text = NULL;
    size_t len = 0;
    if (lm_format_rows(rows, m, ids, lens, b->n, c->row_flags, &text, &len) != LM_OK) die("lm_format_rows failed");
    double t2 = now_s();
    if (len && fwrite(text, 1, len, c->out) != len) die("write failed");
    fflush(c->out); /* outfh.Flush() (search.go:527) */
    double t3 = now_s();
    c->rows += m;
    c->bases += b->bases;
    c->text_bytes += len;
    c->t_search += t1 - t0;
    c->t_format += t2 - t1;
    c->t_write += t3 - t2;
    lm_free(text);
    for (size_t i = 0; i < b->n; i++) {
        free(b->r[i].id);
        free(b->r[i].seq);
    }
    lm_result_free(res); /* RecycleSearchResults */
    free(qs);
    free(ids);
    free(lens);
    b->n = 0;
    b->bases = 0;
    c->flushes++;
}

/* fastx.Reader restricted to what the tests need: FASTA (multi-line) and FASTQ (four-line), plain text; the record id is
 * the header up to the first blank */
static void read_file(ctx *c, batch *b, const char *path, int k, size_t batch_bases, size_t batch_queries) {
    FILE *f = strcmp(path, "-") == 0 ? stdin : fopen(path, "r");
    if (!f) die("cannot open the query file");
    char *line = NULL, *id = NULL, *seq = NULL;
    size_t lcap = 0, slen = 0, scap = 0;
    ssize_t got;
    int fastq = 0, in_qual = 0;
    size_t qual_left = 0;
    for (;;) {
        got = getline(&line, &lcap, f);
        size_t l = got > 0 ? (size_t)got : 0;
        while (l && (line[l - 1] == '\n' || line[l - 1] == '\r')) line[--l] = 0;
        const int eof = got < 0;
        const int header = !eof && !in_qual && (line[0] == '>' || (line[0] == '@' && (id == NULL || fastq)));
        if (eof || header) {
            if (id) { /* a record is complete */
                if ((int)slen < k) {
                    c->total++; /* search.go:571-575: shorter than k: counted, not searched */
                } else {
                    batch_push(b, id, seq ? seq : "", slen);
                    if (b->bases >= batch_bases || b->n >= batch_queries) flush(c, b);
                }
                free(id);
                id = NULL;
            }
            if (eof) break;
            fastq = line[0] == '@';
            char *sp = strpbrk(line + 1, " \t");
            if (sp) *sp = 0;
            id = strdup(line + 1);
            slen = 0;
            continue;
        }
        if (!id) continue; /* text before the first header */
        if (fastq && !in_qual && line[0] == '+') {
            in_qual = 1;
            qual_left = slen;
            if (qual_left == 0) in_qual = 0;
            continue;
        }
        if (in_qual) { /* quality lines: as many characters as bases */
            qual_left = l >= qual_left ? 0 : qual_left - l;
            if (qual_left == 0) in_qual = 0;
            continue;
        }
        if (slen + l + 1 > scap) {
            scap = 2 * (slen + l + 1);
            seq = (char *)realloc(seq, scap);
            if (!seq) die("out of memory");
        }
        for (size_t i = 0; i < l; i++)
            if (!isspace((unsigned char)line[i])) seq[slen++] = line[i];
    }
    free(line);
    free(seq);
    if (f != stdin) fclose(f);
}

static double num(const char *flag, const char *v) {
    char *end = NULL;
    double x = strtod(v, &end);
    if (!end || *end) {
        fprintf(stderr, "[ERRO] invalid value for %s: %s\n", flag, v);
        exit(255);
    }
    return x;
}

int main(int argc, char **argv) {
    lm_options o;
    lm_options_default(&o); /* the defaults of search.go's flag definitions */
    const char *dir = NULL, *out_path = "-";
    int all = 0, show_idx = 0, nfiles = 0;
    size_t batch_bases = (size_t)64 << 20, batch_queries = 4096; /* the batch loop of INTEGRATION.md section 2b */
    const char **files = (const char **)calloc((size_t)argc, sizeof *files);
    if (!files) die("out of memory");
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
#define VAL() (i + 1 < argc ? argv[++i] : (die("flag needs a value"), ""))
        if (!strcmp(a, "-d") || !strcmp(a, "--index")) dir = VAL();
        else if (!strcmp(a, "-o") || !strcmp(a, "--out-file")) out_path = VAL();
        else if (!strcmp(a, "-a") || !strcmp(a, "--all")) all = 1;
        else if (!strcmp(a, "--show-sseq-idx")) show_idx = 1;
        else if (!strcmp(a, "-p") || !strcmp(a, "--seed-min-prefix")) o.min_prefix = (int32_t)num(a, VAL());
        else if (!strcmp(a, "-P") || !strcmp(a, "--seed-min-single-prefix")) o.min_single_prefix = (int32_t)num(a, VAL());
        else if (!strcmp(a, "--seed-max-gap")) o.max_gap = num(a, VAL());
        else if (!strcmp(a, "--seed-max-dist")) o.max_distance = num(a, VAL());
        else if (!strcmp(a, "--align-ext-len")) o.ext_len = (int32_t)num(a, VAL());
        else if (!strcmp(a, "-n") || !strcmp(a, "--top-n-genomes")) o.top_n_genomes = (int32_t)num(a, VAL());
        else if (!strcmp(a, "-N") || !strcmp(a, "--top-n-chains")) o.top_n_chains = (int32_t)num(a, VAL());
        else if (!strcmp(a, "-l") || !strcmp(a, "--align-min-match-len")) o.align_min_match_len = (int32_t)num(a, VAL());
        else if (!strcmp(a, "--align-max-gap")) o.align_max_gap = (int32_t)num(a, VAL());
        else if (!strcmp(a, "--align-band")) o.align_band = (int32_t)num(a, VAL());
        else if (!strcmp(a, "-Q") || !strcmp(a, "--min-qcov-per-genome")) o.min_qcov_per_genome = num(a, VAL());
        else if (!strcmp(a, "-q") || !strcmp(a, "--min-qcov-per-hsp")) o.min_qcov_per_hsp = num(a, VAL());
        else if (!strcmp(a, "-i") || !strcmp(a, "--align-min-match-pident")) o.align_min_pident = num(a, VAL());
        else if (!strcmp(a, "-e") || !strcmp(a, "--max-evalue")) o.max_evalue = num(a, VAL());
        else if (!strcmp(a, "--batch-bases")) batch_bases = (size_t)num(a, VAL());
        else if (!strcmp(a, "--batch-queries")) batch_queries = (size_t)num(a, VAL());
        /* accepted and not passed on: the index is whole in HBM, the batch replaces the goroutines */
        else if (!strcmp(a, "-w") || !strcmp(a, "--load-whole-seeds")) {}
        else if (!strcmp(a, "-J") || !strcmp(a, "--max-query-conc") || !strcmp(a, "--max-open-files") || !strcmp(a, "--gc-interval")) (void)VAL();
        else if (a[0] == '-' && a[1]) { fprintf(stderr, "[ERRO] unknown flag: %s\n", a); return 255; }
        else files[nfiles++] = a;
#undef VAL
    }
    if (!dir && nfiles == 2) { /* the short form: <index dir> <query file> */
        dir = files[0];
        files[0] = files[1];
        nfiles = 1;
    }
    if (!dir) die("flag -d/--index needed"); /* search.go:160-162 */
    /* the checks of search.go:163-227 with the reference's messages; the library repeats the ones whose violation would
     * corrupt a search (LM_ERR_OPTION) */
    char msg[256];
    if (o.min_prefix > 32 || o.min_prefix < 5) {
        snprintf(msg, sizeof msg, "the value of flag -p/--seed-min-prefix (%d) should be in the range of [5, 32]", o.min_prefix);
        die(msg);
    }
    if (o.min_single_prefix > 32) {
        snprintf(msg, sizeof msg, "the value of flag -P/--seed-min-single-prefix (%d) should be <= 32", o.min_single_prefix);
        die(msg);
    }
    if (o.min_single_prefix < o.min_prefix) {
        snprintf(msg, sizeof msg, "the value of flag -P/--seed-min-single-prefix (%d) should be >= that of -p/--seed-min-prefix (%d)",
                 o.min_single_prefix, o.min_prefix);
        die(msg);
    }
    if (o.align_min_match_len < o.min_single_prefix) {
        snprintf(msg, sizeof msg, "the value of flag -l/--align-min-match-len (%d) should be >= that of -M/--seed-min-single-prefix (%d)",
                 o.align_min_match_len, o.min_single_prefix);
        die(msg);
    }
    if (o.align_band < o.align_max_gap) die("the value of flag --align-band should not be smaller thant the value of --align-max-gap");
    if (o.min_qcov_per_genome > 100 || o.min_qcov_per_genome < 0) die("the value of flag -Q/--min-qcov-per-genome should be in range of [0, 100]");
    if (o.align_min_pident < 60 || o.align_min_pident > 100) die("the value of flag -i/--align-min-match-pident should be in range of [60, 100]");
    if (o.min_qcov_per_hsp > 100 || o.min_qcov_per_hsp < 0) die("the value of flag -q/--min-qcov-per-hsp should be in range of [0, 100]");
    if (all) o.output_seq = 1;
    if (nfiles == 0) files[nfiles++] = "-"; /* stdin */

    ctx c;
    memset(&c, 0, sizeof c);
    const double t_open0 = now_s();
    if (lm_index_open(dir, &o, 0, &c.h) != LM_OK) die(lm_last_error(NULL)); /* NewIndexSearcher */
    const double t_open = now_s() - t_open0, t_run0 = now_s();
    lm_index_info info;
    lm_index_get_info(c.h, &info);
    c.out = strcmp(out_path, "-") == 0 ? stdout : fopen(out_path, "w");
    if (!c.out) die("cannot open the output file");
    c.row_flags = (all ? LM_ROW_ALL : 0) | (show_idx ? LM_ROW_SSEQ_IDX : 0);
    fputs(lm_tsv_header(all), c.out); /* search.go:426-430 */
    fputc('\n', c.out);

    batch b;
    memset(&b, 0, sizeof b);
    for (int i = 0; i < nfiles; i++) read_file(&c, &b, files[i], info.k, batch_bases, batch_queries);
    flush(&c, &b);

    /* search.go:616-626 */
    fprintf(stderr, "processed queries: %llu\n", c.total);
    fprintf(stderr, "%.4f%% (%llu/%llu) queries matched\n", c.total ? (double)c.matched / (double)c.total * 100 : 0.0, c.matched, c.total);
    fprintf(stderr, "k=%d masks=%d rows=%llu batches=%d\n", info.k, info.masks, c.rows, c.flushes);
    /* where the host program's time went (bench.py reads this line: host_end_to_end) */
    fprintf(stderr, "timing: open_s=%.3f run_s=%.3f search_s=%.3f format_s=%.3f write_s=%.3f queries=%llu query_bases=%llu rows=%llu tsv_bytes=%llu\n", t_open,
            now_s() - t_run0, c.t_search, c.t_format, c.t_write, c.total, c.bases, c.rows, c.text_bytes);
    if (c.out != stdout) fclose(c.out);
    free(b.r);
    free((void *)files);
    lm_index_close(c.h); /* (*Index).Close */
    return 0;
}
