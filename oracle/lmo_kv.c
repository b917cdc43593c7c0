/* CPU ORACLE (test infrastructure only) — seed k-mer/value files and the in-RAM searcher.
 * Follows util/varint-GB.go, kv/kv-encoding.go, kv/kv-data.go:126-602 (writer), kv/kv-data.go:619-769 (.idx reader),
 * kv/kv-reader.go:762-1021 (ReadDataOfAMaskAsListAndCreateIndex), kv/kv-searcher2.go:105-549 (InMemorySearcher).
 * Pinned by the known-answer test kv/kv-data_test.go:31-361 (restated in tests/test_oracle_formats.py). */
#include "lmo.h"
#include <stdlib.h>
#include <string.h>

/* ---------------- varint-GB (util/varint-GB.go:28-46, 88-113, 153-177) ---------------- */
static int bytelen64(uint64_t n) {
    if (n < 256ull) return 1;
    if (n < 65536ull) return 2;
    if (n < 16777216ull) return 3;
    if (n < 4294967296ull) return 4;
    if (n < 1099511627776ull) return 5;
    if (n < 281474976710656ull) return 6;
    if (n < 72057594037927936ull) return 7;
    return 8;
}

int lmo_put_uint64s(uint8_t *buf, uint64_t v1, uint64_t v2, uint8_t *ctrl) {
    int n = 0;
    int b1 = bytelen64(v1), b2 = bytelen64(v2);
    uint8_t c = (uint8_t)(b1 - 1);
    for (int i = b1 - 1; i >= 0; i--) buf[n++] = (uint8_t)(v1 >> (8 * i));
    c <<= 3;
    c |= (uint8_t)(b2 - 1);
    for (int i = b2 - 1; i >= 0; i--) buf[n++] = (uint8_t)(v2 >> (8 * i));
    *ctrl = c;
    return n;
}

int lmo_get_uint64s(uint8_t ctrl, const uint8_t *buf, int buflen, uint64_t *v1, uint64_t *v2) {
    int b1 = ((ctrl >> 3) & 7) + 1, b2 = (ctrl & 7) + 1;
    if (buflen < b1 + b2) return 0;
    uint64_t a = 0, b = 0;
    int n = 0;
    for (int j = 0; j < b1; j++) a = (a << 8) | buf[n++];
    for (int j = 0; j < b2; j++) b = (b << 8) | buf[n++];
    *v1 = a;
    *v2 = b;
    return n;
}

static void put_be64(uint8_t *b, uint64_t v) {
    for (int i = 0; i < 8; i++) b[i] = (uint8_t)(v >> (56 - 8 * i));
}
static uint64_t get_be64(const uint8_t *b) {
    uint64_t v = 0;
    for (int i = 0; i < 8; i++) v = (v << 8) | b[i];
    return v;
}
/* kv-encoding.go:30-46: 7 low bytes */
static void put_be56(uint8_t *b, uint64_t v) {
    for (int i = 0; i < 7; i++) b[i] = (uint8_t)(v >> (48 - 8 * i));
}
static uint64_t get_be56(const uint8_t *b) {
    uint64_t v = 0;
    for (int i = 0; i < 7; i++) v = (v << 8) | b[i];
    return v;
}

static inline uint64_t anchor_of(uint64_t kmer, int k, int mp, int ap) { /* kv-data.go:319-325 */
    int shift = (k - mp - ap) << 1;
    uint64_t mask = ((uint64_t)1 << (ap << 1)) - 1;
    return (kmer >> shift) & mask;
}

/* ---------------- writer (kv-data.go:126-602) ---------------- */
static void write_vals(FILE *w, const uint64_t *v, int n, int use7, int64_t *N) {
    uint8_t b[8];
    for (int i = 0; i < n; i++) {
        if (use7) {
            put_be56(b, v[i]);
            fwrite(b, 1, 7, w);
            *N += 7;
        } else {
            put_be64(b, v[i]);
            fwrite(b, 1, 8, w);
            *N += 8;
        }
    }
}

int lmo_kv_write(const char *file, int k, int mask_offset, int nmasks, lmo_kv_rec **recs, const int *nrecs,
                 int mask_prefix, int anchor_prefix, int nbatches) {
    if (mask_prefix + anchor_prefix > k || anchor_prefix == 0) return -1;
    int use7 = nbatches <= 512; /* kv-data.go:137 */
    char fidx[4096];
    snprintf(fidx, sizeof fidx, "%s.idx", file);
    FILE *w = fopen(file, "wb");
    FILE *wi = fopen(fidx, "wb");
    if (!w || !wi) return -1;
    int64_t N = 0;
    uint8_t hdr[32];
    memset(hdr, 0, 32);
    memcpy(hdr, ".kv-data", 8);
    hdr[8] = 1; /* MainVersion */
    hdr[9] = 1; /* MinorVersion */
    hdr[10] = (uint8_t)k;
    hdr[11] = use7 ? 1 : 0;
    put_be64(hdr + 16, (uint64_t)mask_offset);
    put_be64(hdr + 24, (uint64_t)nmasks);
    fwrite(hdr, 1, 32, w);
    N += 32;
    memset(hdr, 0, 32);
    memcpy(hdr, ".kvindex", 8);
    hdr[8] = 1;
    hdr[9] = 1;
    hdr[10] = (uint8_t)k;
    hdr[11] = (uint8_t)mask_prefix;
    hdr[12] = (uint8_t)anchor_prefix;
    hdr[13] = use7 ? 1 : 0;
    put_be64(hdr + 16, (uint64_t)mask_offset);
    put_be64(hdr + 24, (uint64_t)nmasks);
    fwrite(hdr, 1, 32, wi);

    size_t np2o = 2 + ((size_t)1 << (anchor_prefix << 1)) * 2;
    uint64_t *p2o = (uint64_t *)malloc(sizeof(uint64_t) * np2o);
    uint8_t b8[8], bufvar[16], buf[40];

    for (int im = 0; im < nmasks; im++) {
        const lmo_kv_rec *m = recs[im];
        int nk = nrecs[im];
        put_be64(b8, (uint64_t)nk);
        fwrite(b8, 1, 8, w);
        N += 8;
        if (nk == 0) {
            put_be64(b8, 0);
            fwrite(b8, 1, 8, wi);
            continue;
        }
        memset(p2o, 0, sizeof(uint64_t) * np2o);
        p2o[1] = (uint64_t)N << 1;
        int even = (nk & 1) == 0, nm1 = nk - 1;
        int has_prev = 0, first = 1;
        uint64_t pre_key = 0, offset = 0, prefix, prefix_pre = 0;
        const lmo_kv_rec *pre = NULL;
        for (int i = 0; i < nk; i++) {
            uint64_t key = m[i].kmer;
            if (!has_prev) {
                pre_key = key;
                pre = &m[i];
                has_prev = 1;
                continue;
            }
            prefix = anchor_of(pre_key, k, mask_prefix, anchor_prefix);
            if (first || prefix != prefix_pre) {
                first = 0;
                size_t j = (size_t)(prefix << 1) + 2;
                p2o[j] = pre_key;
                p2o[j + 1] = (uint64_t)N << 1;
                prefix_pre = prefix;
            }
            prefix = anchor_of(key, k, mask_prefix, anchor_prefix);
            if (prefix != prefix_pre) {
                size_t j = (size_t)(prefix << 1) + 2;
                p2o[j] = key;
                p2o[j + 1] = ((uint64_t)N << 1) | 1;
                prefix_pre = prefix;
            }
            uint8_t ck, cv;
            int nbk = lmo_put_uint64s(bufvar, pre_key - offset, key - pre_key, &ck);
            if (even && i == nm1) ck |= 1 << 7;
            buf[0] = ck;
            memcpy(buf + 1, bufvar, nbk);
            int n = nbk + 1;
            int nbv = lmo_put_uint64s(bufvar, (uint64_t)pre->nvals, (uint64_t)m[i].nvals, &cv);
            buf[n] = cv;
            memcpy(buf + n + 1, bufvar, nbv);
            n += nbv + 1;
            fwrite(buf, 1, n, w);
            N += n;
            write_vals(w, pre->vals, pre->nvals, use7, &N);
            write_vals(w, m[i].vals, m[i].nvals, use7, &N);
            offset = key;
            has_prev = 0;
        }
        if (has_prev) {
            prefix = anchor_of(pre_key, k, mask_prefix, anchor_prefix);
            if (first || prefix != prefix_pre) {
                first = 0;
                size_t j = (size_t)(prefix << 1) + 2;
                p2o[j] = pre_key;
                p2o[j + 1] = (uint64_t)N << 1;
                prefix_pre = prefix;
            }
            uint8_t ck, cv;
            int nbk = lmo_put_uint64s(bufvar, pre_key - offset, 0, &ck);
            ck |= 1 << 7;
            ck |= 1 << 6;
            buf[0] = ck;
            memcpy(buf + 1, bufvar, nbk);
            int n = nbk + 1;
            int nbv = lmo_put_uint64s(bufvar, (uint64_t)pre->nvals, 0, &cv);
            buf[n] = cv;
            memcpy(buf + n + 1, bufvar, nbv);
            n += nbv + 1;
            fwrite(buf, 1, n, w);
            N += n;
            write_vals(w, pre->vals, pre->nvals, use7, &N);
        }
        /* index records: those with offset > 0 (kv-data.go:566-598) */
        uint64_t nrec = 0;
        size_t e = np2o >> 1;
        for (size_t i = 0; i < e; i++)
            if (p2o[2 * i + 1] > 0) nrec++;
        put_be64(b8, nrec);
        fwrite(b8, 1, 8, wi);
        p2o[0] = nrec;
        for (size_t i = 0; i < e; i++) {
            if (p2o[2 * i + 1] > 0) {
                put_be64(b8, p2o[2 * i]);
                fwrite(b8, 1, 8, wi);
                put_be64(b8, p2o[2 * i + 1]);
                fwrite(b8, 1, 8, wi);
            }
        }
    }
    free(p2o);
    fclose(w);
    fclose(wi);
    return 0;
}

/* ---------------- .idx reader (kv-data.go:619-769): dense [2+2*4^a] per mask ---------------- */
int lmo_kv_read_index(const char *file, int *k, int *chunk_index, int *chunk_size, int *mask_prefix, int *anchor_prefix,
                      uint64_t ***tables) {
    FILE *f = fopen(file, "rb");
    if (!f) return -1;
    uint8_t hdr[32];
    if (fread(hdr, 1, 32, f) != 32 || memcmp(hdr, ".kvindex", 8) || hdr[8] != 1) {
        fclose(f);
        return -2;
    }
    *k = hdr[10];
    *mask_prefix = hdr[11];
    *anchor_prefix = hdr[12];
    *chunk_index = (int)get_be64(hdr + 16);
    *chunk_size = (int)get_be64(hdr + 24);
    int K = *k, mp = *mask_prefix, ap = *anchor_prefix;
    size_t np2o = 2 + ((size_t)1 << (ap << 1)) * 2;
    uint64_t **T = (uint64_t **)calloc(*chunk_size, sizeof(uint64_t *));
    uint8_t b[16];
    for (int i = 0; i < *chunk_size; i++) {
        if (fread(b, 1, 8, f) != 8) break;
        uint64_t nrec = get_be64(b);
        if (nrec == 0) continue;
        uint64_t *t = (uint64_t *)calloc(np2o, sizeof(uint64_t));
        for (uint64_t r = 0; r < nrec; r++) {
            if (fread(b, 1, 16, f) != 16) break;
            uint64_t kmer = get_be64(b), off = get_be64(b + 8);
            if (r == 0) {
                t[0] = kmer;
                t[1] = off;
            } else {
                size_t j = (size_t)(anchor_of(kmer, K, mp, ap) << 1) + 2;
                t[j] = kmer;
                t[j + 1] = off;
            }
        }
        T[i] = t;
    }
    fclose(f);
    *tables = T;
    return 0;
}

/* ---------------- in-RAM loader (kv-searcher2.go:52-88 -> kv-reader.go:762-1021) ---------------- */
lmo_kv_mem *lmo_kv_load(const char *file) {
    char fidx[4096];
    snprintf(fidx, sizeof fidx, "%s.idx", file);
    FILE *fi = fopen(fidx, "rb");
    if (!fi) return NULL;
    uint8_t ih[32];
    if (fread(ih, 1, 32, fi) != 32 || memcmp(ih, ".kvindex", 8)) {
        fclose(fi);
        return NULL;
    }
    fclose(fi);
    FILE *f = fopen(file, "rb");
    if (!f) return NULL;
    uint8_t hdr[32];
    if (fread(hdr, 1, 32, f) != 32 || memcmp(hdr, ".kv-data", 8) || hdr[8] != 1) {
        fclose(f);
        return NULL;
    }
    lmo_kv_mem *m = (lmo_kv_mem *)calloc(1, sizeof *m);
    m->k = hdr[10];
    m->use7 = hdr[11] & 1;
    m->chunk_index = (int)get_be64(hdr + 16);
    m->chunk_size = (int)get_be64(hdr + 24);
    m->mask_prefix = ih[11];
    m->anchor_prefix = ih[12];
    int K = m->k, mp = m->mask_prefix, ap = m->anchor_prefix;
    size_t nidx = (size_t)1 << (ap << 1);
    m->kv = (uint64_t **)calloc(m->chunk_size, sizeof(uint64_t *));
    m->kvlen = (int64_t *)calloc(m->chunk_size, sizeof(int64_t));
    m->index = (int64_t **)calloc(m->chunk_size, sizeof(int64_t *));
    int nvb = m->use7 ? 7 : 8;
    uint8_t b[40];
    for (int im = 0; im < m->chunk_size; im++) {
        if (fread(b, 1, 8, f) != 8) goto broken;
        int64_t nk = (int64_t)get_be64(b);
        if (nk == 0) continue;
        int64_t cap = nk * 2 + 16, len = 0;
        uint64_t *d = (uint64_t *)malloc(sizeof(uint64_t) * cap);
        int64_t *index = (int64_t *)malloc(sizeof(int64_t) * nidx);
        for (size_t i = 0; i < nidx; i++) index[i] = -1;
        uint64_t off = 0, prefix, prefix_pre = 0;
        int first = 1;
        for (;;) {
            if (fread(b, 1, 1, f) != 1) goto broken;
            uint8_t ctrl = b[0];
            int last_pair = (ctrl & 128) > 0, has2 = (ctrl & 64) == 0;
            ctrl &= 63;
            int nb = ((ctrl >> 3) & 7) + (ctrl & 7) + 2;
            if ((int)fread(b, 1, nb, f) != nb) goto broken;
            uint64_t v1, v2;
            lmo_get_uint64s(ctrl, b, nb, &v1, &v2);
            uint64_t kmer1 = v1 + off, kmer2 = kmer1 + v2;
            off = kmer2;
            prefix = anchor_of(kmer1, K, mp, ap);
            if (first || prefix != prefix_pre) {
                first = 0;
                index[prefix] = len;
                prefix_pre = prefix;
            }
            if (fread(b, 1, 1, f) != 1) goto broken;
            ctrl = b[0];
            nb = ((ctrl >> 3) & 7) + (ctrl & 7) + 2;
            if ((int)fread(b, 1, nb, f) != nb) goto broken;
            uint64_t l1, l2;
            lmo_get_uint64s(ctrl, b, nb, &l1, &l2);
            for (int which = 0; which < 2; which++) {
                uint64_t kmer = which == 0 ? kmer1 : kmer2;
                uint64_t lv = which == 0 ? l1 : l2;
                if (which == 1) {
                    if (last_pair && !has2) break;
                    prefix = anchor_of(kmer2, K, mp, ap);
                    if (prefix != prefix_pre) {
                        index[prefix] = len;
                        prefix_pre = prefix;
                    }
                }
                if (len + (int64_t)lv * 2 > cap) {
                    cap = (len + (int64_t)lv * 2) * 2;
                    d = (uint64_t *)realloc(d, sizeof(uint64_t) * cap);
                }
                for (uint64_t j = 0; j < lv; j++) {
                    if ((int)fread(b, 1, nvb, f) != nvb) goto broken;
                    d[len++] = kmer;
                    d[len++] = m->use7 ? get_be56(b) : get_be64(b);
                }
            }
            if (last_pair) break;
        }
        m->kv[im] = d;
        m->kvlen[im] = len;
        m->index[im] = index;
    }
    fclose(f);
    return m;
broken:
    fclose(f);
    lmo_kv_free(m);
    return NULL;
}

void lmo_kv_free(lmo_kv_mem *m) {
    if (!m) return;
    for (int i = 0; i < m->chunk_size; i++) {
        free(m->kv[i]);
        free(m->index[i]);
    }
    free(m->kv);
    free(m->kvlen);
    free(m->index);
    free(m);
}

/* ---------------- searcher (kv-searcher2.go:105-323 / 326-549) ---------------- */
void lmo_kv_results_free(lmo_kv_results *r) {
    free(r->sr);
    free(r->vals);
    memset(r, 0, sizeof *r);
}
static lmo_kv_sr *push_sr(lmo_kv_results *r) {
    if (r->n == r->cap) {
        r->cap = r->cap ? r->cap * 2 : 64;
        r->sr = (lmo_kv_sr *)realloc(r->sr, sizeof(lmo_kv_sr) * r->cap);
    }
    return &r->sr[r->n++];
}
static void push_val(lmo_kv_results *r, uint64_t v) {
    if (r->nv == r->capv) {
        r->capv = r->capv ? r->capv * 2 : 256;
        r->vals = (uint64_t *)realloc(r->vals, sizeof(uint64_t) * r->capv);
    }
    r->vals[r->nv++] = v;
}

/* one (mask, kmer) probe: the body shared by Search and Search2 */
static void search_one(const lmo_kv_mem *m, int iQ, int iKmer, uint64_t kmer, int p, int check_flag, int reversed,
                       lmo_kv_results *out) {
    int k = m->k;
    const uint64_t *data = m->kv[iQ];
    int64_t ndata = m->kvlen[iQ];
    if (ndata == 0 || kmer == 0) return;
    int64_t last = ndata - 2;
    const int64_t *index = m->index[iQ];
    uint64_t rvflag = reversed ? 1 : 0;
    uint64_t left, right;
    if (p < k) {
        int suffix2 = (k - p) << 1;
        uint64_t mask = ((uint64_t)1 << suffix2) - 1;
        left = kmer & (~(uint64_t)0 - mask);
        right = ((kmer >> suffix2) << suffix2) + mask;
    } else {
        left = right = kmer;
    }
    uint64_t anchor = anchor_of(left, k, m->mask_prefix, m->anchor_prefix);
    int64_t i = index[anchor];
    if (i < 0) return;
    uint64_t last_next = (uint64_t)1 << (m->anchor_prefix << 1);
    uint64_t anchor_next = anchor;
    int64_t j = -1;
    while (j < 0 && anchor_next + 1 < last_next) {
        anchor_next++;
        j = index[anchor_next];
    }
    if (j > 0 && j - i > 2) {
        int64_t begin = i, end = j, middle;
        for (;;) {
            middle = begin + ((end - begin) >> 1);
            if (middle & 1) middle--;
            if (middle == begin) {
                i = begin;
                break;
            }
            if (left <= data[middle])
                end = middle;
            else
                begin = middle;
            if (begin + 2 == end) {
                i = begin;
                break;
            }
        }
    }
    int found = 0, first = 1;
    int cur = -1; /* index of current result in out, -1 = nil */
    uint64_t kmer0 = 0;
    for (;;) {
        uint64_t kmer1 = data[i];
        if (kmer1 > right) break;
        if (kmer1 >= left) found = 1;
        if (found) {
            if (kmer1 != kmer0 || first) {
                lmo_kv_sr *s = push_sr(out);
                cur = out->n - 1;
                s->iquery = iQ + m->chunk_index;
                s->iquery2 = iKmer;
                {
                    uint64_t x = kmer ^ kmer1; /* bits.LeadingZeros64(0) == 64 */
                    int lz = x ? __builtin_clzll(x) : 64;
                    s->len = (uint8_t)((lz >> 1) + k - 32);
                }
                s->is_suffix = (uint8_t)reversed;
                s->val_off = out->nv;
                s->nvals = 0;
                first = 0;
            }
            if (!check_flag || (data[i + 1] & 1) == rvflag) {
                push_val(out, data[i + 1]);
                out->sr[cur].nvals++;
            }
            kmer0 = kmer1;
        } else {
            cur = -1;
        }
        if (i == last) break;
        i += 2;
    }
}

int lmo_kv_search(const lmo_kv_mem *m, const uint64_t *kmers, int p, int check_flag, int reversed, lmo_kv_results *out) {
    if (p < m->mask_prefix + m->anchor_prefix || p > m->k) return -1;
    for (int iQ = 0; iQ < m->chunk_size; iQ++) search_one(m, iQ, 0, kmers[iQ], p, check_flag, reversed, out);
    return 0;
}

int lmo_kv_search2(const lmo_kv_mem *m, const uint64_t *kmersR, const int *kr_off, int p, int check_flag, int reversed,
                   lmo_kv_results *out) {
    if (p < m->mask_prefix + m->anchor_prefix || p > m->k) return -1;
    for (int iQ = 0; iQ < m->chunk_size; iQ++)
        for (int j = kr_off[iQ]; j < kr_off[iQ + 1]; j++)
            search_one(m, iQ, j - kr_off[iQ], kmersR[j], p, check_flag, reversed, out);
    return 0;
}
