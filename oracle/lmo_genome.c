/* CPU ORACLE (test infrastructure only) — 2-bit genome container.
 * Follows genome/genome.go:184-358 (writer), :388-474 (reader/index), :931-1143 (SubSeq3), :1427-1550 (codec).
 * Pinned by genome/genome_test.go:30-164 (restated in tests/test_oracle_formats.py). */
#include "lmo.h"
#include <stdlib.h>
#include <string.h>

static void put_be16(uint8_t *b, uint32_t v) {
    b[0] = (uint8_t)(v >> 8);
    b[1] = (uint8_t)v;
}
static void put_be32(uint8_t *b, uint32_t v) {
    b[0] = (uint8_t)(v >> 24);
    b[1] = (uint8_t)(v >> 16);
    b[2] = (uint8_t)(v >> 8);
    b[3] = (uint8_t)v;
}
static void put_be64(uint8_t *b, uint64_t v) {
    for (int i = 0; i < 8; i++) b[i] = (uint8_t)(v >> (56 - 8 * i));
}
static uint32_t get_be16(const uint8_t *b) { return ((uint32_t)b[0] << 8) | b[1]; }
static uint32_t get_be32(const uint8_t *b) {
    return ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3];
}
static uint64_t get_be64(const uint8_t *b) {
    uint64_t v = 0;
    for (int i = 0; i < 8; i++) v = (v << 8) | b[i];
    return v;
}

/* genome.go:1471-1508: first base in bits 7-6 */
int lmo_seq2twobit(const uint8_t *s, int len, uint8_t *out) {
    int n = len >> 2, m = len & 3;
    for (int i = 0; i < n; i++) {
        int j = i << 2;
        out[i] = (uint8_t)((lmo_base2bit[s[j]] << 6) + (lmo_base2bit[s[j + 1]] << 4) + (lmo_base2bit[s[j + 2]] << 2) +
                           lmo_base2bit[s[j + 3]]);
    }
    if (m == 0) return n;
    int j = n << 2;
    uint8_t b = 0;
    if (m >= 1) b |= (uint8_t)(lmo_base2bit[s[j]] << 6);
    if (m >= 2) b |= (uint8_t)(lmo_base2bit[s[j + 1]] << 4);
    if (m >= 3) b |= (uint8_t)(lmo_base2bit[s[j + 2]] << 2);
    out[n] = b;
    return n + 1;
}

static const char bit2base[4] = {'A', 'C', 'G', 'T'};

void lmo_twobit2seq(const uint8_t *b2, int bases, uint8_t *out) {
    for (int i = 0; i < bases; i++) out[i] = (uint8_t)bit2base[(b2[i >> 2] >> ((3 - (i & 3)) << 1)) & 3];
}

struct lmo_gwriter {
    FILE *fh;
    char *file;
    uint32_t batch;
    int64_t offset;
    int64_t *idx; /* pairs offset,len */
    int n, cap;
};

lmo_gwriter *lmo_gwriter_open(const char *file, uint32_t batch) {
    lmo_gwriter *w = (lmo_gwriter *)calloc(1, sizeof *w);
    w->fh = fopen(file, "wb");
    if (!w->fh) {
        free(w);
        return NULL;
    }
    w->file = strdup(file);
    w->batch = batch;
    uint8_t hdr[16];
    memset(hdr, 0, 16);
    memcpy(hdr, ".genomes", 8);
    hdr[8] = 0; /* MainVersion */
    hdr[9] = 1; /* MinorVersion */
    fwrite(hdr, 1, 16, w->fh);
    w->offset = 16;
    return w;
}

/* genome.go:217-306 */
int lmo_gwriter_write(lmo_gwriter *w, const lmo_genome_in *g) {
    if (w->n == w->cap) {
        w->cap = w->cap ? w->cap * 2 : 64;
        w->idx = (int64_t *)realloc(w->idx, sizeof(int64_t) * 2 * w->cap);
    }
    w->idx[2 * w->n] = w->offset;
    w->idx[2 * w->n + 1] = g->len;
    w->n++;
    uint8_t b[16];
    int64_t nw = 0;
    size_t idlen = strlen(g->id);
    if (idlen > 65535) idlen = 65535;
    put_be16(b, (uint32_t)idlen);
    fwrite(b, 1, 2, w->fh);
    fwrite(g->id, 1, idlen, w->fh);
    nw += 2 + (int64_t)idlen;
    put_be32(b, (uint32_t)g->genome_size);
    put_be32(b + 4, (uint32_t)g->len);
    put_be32(b + 8, (uint32_t)g->nseqs);
    fwrite(b, 1, 12, w->fh);
    nw += 12;
    for (int i = 0; i < g->nseqs; i++) {
        put_be32(b, (uint32_t)g->seq_sizes[i]);
        size_t l = strlen(g->seq_ids[i]);
        if (l > 65535) l = 65535;
        put_be16(b + 4, (uint32_t)l);
        fwrite(b, 1, 6, w->fh);
        fwrite(g->seq_ids[i], 1, l, w->fh);
        nw += 6 + (int64_t)l;
    }
    int nbytes = (g->len + 3) >> 2;
    uint8_t *b2 = (uint8_t *)malloc(nbytes > 0 ? nbytes : 1);
    int nb = lmo_seq2twobit(g->seq, g->len, b2);
    put_be32(b, (uint32_t)nb);
    put_be32(b + 4, (uint32_t)g->len);
    fwrite(b, 1, 8, w->fh);
    fwrite(b2, 1, nb, w->fh);
    nw += 8 + nb;
    free(b2);
    w->offset += nw;
    return 0;
}

/* genome.go:309-357 */
int lmo_gwriter_close(lmo_gwriter *w) {
    fclose(w->fh);
    char fidx[4096];
    snprintf(fidx, sizeof fidx, "%s.idx", w->file);
    FILE *f = fopen(fidx, "wb");
    if (!f) return -1;
    uint8_t hdr[16], b[12];
    memset(hdr, 0, 16);
    memcpy(hdr, ".genomei", 8);
    hdr[8] = 0;
    hdr[9] = 1;
    fwrite(hdr, 1, 16, f);
    put_be32(b, w->batch);
    put_be32(b + 4, (uint32_t)w->n);
    fwrite(b, 1, 8, f);
    for (int i = 0; i < w->n; i++) {
        put_be64(b, (uint64_t)w->idx[2 * i]);
        put_be32(b + 8, (uint32_t)w->idx[2 * i + 1]);
        fwrite(b, 1, 12, f);
    }
    fclose(f);
    free(w->idx);
    free(w->file);
    free(w);
    return 0;
}

/* genome.go:388-474 */
lmo_greader *lmo_greader_open(const char *file) {
    char fidx[4096];
    snprintf(fidx, sizeof fidx, "%s.idx", file);
    FILE *fi = fopen(fidx, "rb");
    if (!fi) return NULL;
    uint8_t hdr[16], b[12];
    if (fread(hdr, 1, 16, fi) != 16 || memcmp(hdr, ".genomei", 8) || hdr[8] != 0) {
        fclose(fi);
        return NULL;
    }
    if (fread(b, 1, 8, fi) != 8) {
        fclose(fi);
        return NULL;
    }
    lmo_greader *r = (lmo_greader *)calloc(1, sizeof *r);
    r->batch = get_be32(b);
    r->n = get_be32(b + 4);
    r->offsets = (uint64_t *)malloc(sizeof(uint64_t) * (r->n ? r->n : 1));
    r->bases = (uint32_t *)malloc(sizeof(uint32_t) * (r->n ? r->n : 1));
    for (uint32_t i = 0; i < r->n; i++) {
        if (fread(b, 1, 12, fi) != 12) break;
        r->offsets[i] = get_be64(b);
        r->bases[i] = get_be32(b + 8);
    }
    fclose(fi);
    r->fh = fopen(file, "rb");
    if (!r->fh || fread(hdr, 1, 16, r->fh) != 16 || memcmp(hdr, ".genomes", 8) || hdr[8] != 0) {
        lmo_greader_close(r);
        return NULL;
    }
    return r;
}

void lmo_greader_close(lmo_greader *r) {
    if (!r) return;
    if (r->fh) fclose(r->fh);
    free(r->offsets);
    free(r->bases);
    free(r);
}

void lmo_genome_free(lmo_genome *g) {
    if (!g) return;
    for (int i = 0; i < g->nseqs; i++) free(g->seq_ids[i]);
    free(g->seq_ids);
    free(g->seq_sizes);
    free(g->seq);
    free(g);
}

/* genome.go:931-1143 */
lmo_genome *lmo_subseq3(lmo_greader *r, int idx, int start, int end, lmo_genome *g) {
    if (idx < 0 || idx >= (int)r->n) return NULL;
    int nbases = (int)r->bases[idx];
    if (start < 0) start = 0;
    if (end >= nbases - 1) end = nbases - 1;
    if (end < start) end = start;
    int64_t offset;
    uint8_t b[16];
    if (g == NULL) {
        offset = (int64_t)r->offsets[idx];
        g = (lmo_genome *)calloc(1, sizeof *g);
        fseek(r->fh, offset, SEEK_SET);
        if (fread(b, 1, 2, r->fh) != 2) goto broken;
        uint32_t idlen = get_be16(b);
        offset += 2;
        fseek(r->fh, idlen, SEEK_CUR);
        offset += idlen;
        if (fread(b, 1, 12, r->fh) != 12) goto broken;
        g->genome_size = (int)get_be32(b);
        g->len = (int)get_be32(b + 4);
        g->nseqs = (int)get_be32(b + 8);
        offset += 12;
        g->seq_sizes = (int *)malloc(sizeof(int) * (g->nseqs ? g->nseqs : 1));
        g->seq_ids = (char **)calloc(g->nseqs ? g->nseqs : 1, sizeof(char *));
        for (int i = 0; i < g->nseqs; i++) {
            if (fread(b, 1, 6, r->fh) != 6) goto broken;
            g->seq_sizes[i] = (int)get_be32(b);
            int l = (int)get_be16(b + 4);
            g->seq_ids[i] = (char *)malloc(l + 1);
            if ((int)fread(g->seq_ids[i], 1, l, r->fh) != l) goto broken;
            g->seq_ids[i][l] = 0;
            offset += 6 + l;
        }
        g->seq_offset = offset;
    } else {
        offset = g->seq_offset;
    }
    offset += 8 + (int64_t)(start >> 2);
    fseek(r->fh, offset, SEEK_SET);
    int nbytes = (end >> 2) - (start >> 2) + 1;
    uint8_t *buf = (uint8_t *)malloc(nbytes);
    if ((int)fread(buf, 1, nbytes, r->fh) < nbytes) {
        free(buf);
        goto broken;
    }
    int l = end - start + 1;
    if (g->seqcap < l + 8) {
        g->seqcap = l + 8;
        g->seq = (uint8_t *)realloc(g->seq, g->seqcap);
    }
    /* equivalent to the byte-wise first/middle/last decode of genome.go:1060-1139 */
    for (int i = 0; i < l; i++) {
        int p = (start & 3) + i;
        g->seq[i] = (uint8_t)bit2base[(buf[p >> 2] >> ((3 - (p & 3)) << 1)) & 3];
    }
    free(buf);
    g->seqlen = l;
    return g;
broken:
    lmo_genome_free(g);
    return NULL;
}
