/* tests/cabi_shim.c — the call sequence of the cgo shim of INTEGRATION.md, in plain C99.
 *
 * A Go toolchain is not available here, so the Go file cannot be compiled; what CAN be checked is everything cgo relies
 * on: include/lexicmap_hip.h is valid C (cgo compiles the preamble with a C compiler, not C++), every call of the shim
 * links against liblexicmap_hip.so with C linkage, and the sequence open -> search_batch -> rows -> format_row -> free ->
 * close produces the TSV lines.  tests/test_cabi_and_host.py compiles and links this file on the CPU box (no run);
 * tests/test_gpu_parity.py runs it on the GPU and compares its output with the ctypes path.
 *
 *   usage: cabi_shim <index dir> <fasta>      (prints one TSV line per HSP row, default columns)
 */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>

#include "lexicmap_hip.h"

int main(int argc, char **argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s <index dir> <fasta>\n", argv[0]);
        return 2;
    }
    /* NewHipIndexSearcher */
    lm_options o;
    lm_options_default(&o);
    lm_index *h = NULL;
    if (lm_index_open(argv[1], &o, 0, &h) != LM_OK) {
        fprintf(stderr, "lexicmap_hip: %s\n", lm_last_error(NULL));
        return 1;
    }
    lm_index_info info;
    lm_index_get_info(h, &info);
    /* reader loop of search.go:548-608: records upper-cased, one batch */
    FILE *f = fopen(argv[2], "r");
    if (!f) return 2;
    size_t cap = 16, n = 0;
    char **ids = (char **)malloc(cap * sizeof *ids);
    char **seqs = (char **)malloc(cap * sizeof *seqs);
    size_t *lens = (size_t *)malloc(cap * sizeof *lens);
    char line[1 << 16];
    while (fgets(line, sizeof line, f)) {
        size_t l = strlen(line);
        while (l && (line[l - 1] == '\n' || line[l - 1] == '\r')) line[--l] = 0;
        if (line[0] == '>') {
            if (n == cap) {
                cap *= 2;
                ids = (char **)realloc(ids, cap * sizeof *ids);
                seqs = (char **)realloc(seqs, cap * sizeof *seqs);
                lens = (size_t *)realloc(lens, cap * sizeof *lens);
            }
            char *sp = strpbrk(line + 1, " \t");
            if (sp) *sp = 0;
            ids[n] = strdup(line + 1);
            seqs[n] = (char *)calloc(1, 1);
            lens[n] = 0;
            n++;
        } else if (n) {
            seqs[n - 1] = (char *)realloc(seqs[n - 1], lens[n - 1] + l + 1);
            for (size_t i = 0; i < l; i++) seqs[n - 1][lens[n - 1] + i] = (char)toupper((unsigned char)line[i]);
            lens[n - 1] += l;
        }
    }
    fclose(f);
    /* SearchBatch */
    lm_query *qs = (lm_query *)malloc((n ? n : 1) * sizeof *qs);
    for (size_t i = 0; i < n; i++) {
        qs[i].seq = (const uint8_t *)seqs[i];
        qs[i].len = (uint32_t)lens[i];
    }
    lm_result *res = NULL;
    if (lm_search_batch(h, qs, n, &res) != LM_OK) {
        fprintf(stderr, "lexicmap_hip: %s\n", lm_last_error(h));
        return 1;
    }
    const lm_hsp *rows = NULL;
    size_t m = lm_result_rows(res, &rows);
    static char buf[1 << 16];
    for (size_t i = 0; i < m; i++) { /* the printer of search.go:506-516 */
        lm_format_row(&rows[i], ids[rows[i].query], (uint32_t)lens[rows[i].query], 0, buf, sizeof buf);
        puts(buf);
    }
    lm_stage_stats st;
    lm_result_stats(res, &st);
    fprintf(stderr, "k=%d masks=%d queries=%zu rows=%zu\n", info.k, info.masks, n, m);
    lm_result_free(res); /* RecycleSearchResults */
    lm_index_close(h);   /* (*Index).Close */
    return 0;
}
