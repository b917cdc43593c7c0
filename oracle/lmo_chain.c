/* CPU ORACLE (test infrastructure only) — anchor de-duplication and the three chainers.
 * Follows lib-index-search.go:864-990 (ClearSubstrPairs), lib-chaining.go:122-667 (Chainer),
 * rangeindex/range_index.go:68-121, lib-chaining2.go:152-658 (Chainer2 + chainARegion),
 * lib-chaining3.go:111-299 (Chainer3), lib-seq_compare.go:553-634 (TrimSubStrPairs, overlap).
 *
 * Determinism notes:
 *  - slices.SortFunc is unstable and the reference's anchor arrival order is goroutine-dependent; anchors that tie
 *    on (QBegin, QEnd, TBegin) differ only in strand flags.  The oracle (and the HIP path) use the total order
 *    (QBegin asc, QEnd desc, TBegin asc, QRC asc, TRC asc).
 *  - float32 arithmetic is evaluated operation by operation (compile with -ffp-contract=off); gapScore's log2 follows
 *    Go's pure-Go math.Log2/math.Log (FreeBSD e_log.c) so the float64->float32 rounding is the same.
 */
#include "lmo.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <limits.h>

/* ---------------------------------------------------------------------------------------------
 * sort + ClearSubstrPairs */
static int cmp_sub(const void *pa, const void *pb) {
    const lmo_sub *a = (const lmo_sub *)pa, *b = (const lmo_sub *)pb;
    if (a->qbegin != b->qbegin) return a->qbegin < b->qbegin ? -1 : 1;
    int ae = a->qbegin + a->len, be = b->qbegin + b->len;
    if (ae != be) return be < ae ? -1 : 1; /* QEnd descending */
    if (a->tbegin != b->tbegin) return a->tbegin < b->tbegin ? -1 : 1;
    if (a->qrc != b->qrc) return a->qrc < b->qrc ? -1 : 1;
    if (a->trc != b->trc) return a->trc < b->trc ? -1 : 1;
    return 0;
}

void lmo_sort_subs(lmo_sub *subs, int n) { qsort(subs, n, sizeof(lmo_sub), cmp_sub); }

int lmo_clear_subs(lmo_sub *subs, int n, int k) {
    if (n <= 0) return n;
    lmo_sort_subs(subs, n);
    uint8_t *marker = (uint8_t *)calloc(n, 1);
    for (int i = 0; i + 1 < n; i++) { /* i is the index in (*subs)[1:], v = subs[i+1] */
        const lmo_sub *v = &subs[i + 1];
        int32_t vqend = v->qbegin + v->len;
        int32_t upbound = vqend - k;
        if (upbound < 0) upbound = 0;
        int32_t vtbegin = v->tbegin, vtend = v->tbegin + v->len;
        /* first index in subs[0..i] with QBegin >= upbound */
        int lo = 0, hi = i + 1;
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if (subs[mid].qbegin < upbound)
                lo = mid + 1;
            else
                hi = mid;
        }
        for (int j = lo; j <= i; j++) {
            const lmo_sub *p = &subs[j];
            if (vqend <= p->qbegin + p->len && vtbegin >= p->tbegin && vtend <= p->tbegin + p->len) {
                marker[i + 1] = 1;
                break;
            }
        }
    }
    int j = 0;
    for (int i = 0; i < n; i++)
        if (!marker[i]) subs[j++] = subs[i];
    free(marker);
    return j;
}

/* ---------------------------------------------------------------------------------------------
 * Chainer (lib-chaining.go) */

/* Go math.log (pure Go, FreeBSD e_log.c) */
static double go_log(double x) {
    static const double Ln2Hi = 6.93147180369123816490e-01, Ln2Lo = 1.90821492927058770002e-10,
                        L1 = 6.666666666666735130e-01, L2 = 3.999999999940941908e-01, L3 = 2.857142874366239149e-01,
                        L4 = 2.222219843214978396e-01, L5 = 1.818357216161805012e-01, L6 = 1.531383769920937332e-01,
                        L7 = 1.479819860511658591e-01;
    int ki;
    double f1 = frexp(x, &ki);
    if (f1 < 1.4142135623730950488016887242096980785696718753769480731766797379907324784621 / 2) {
        f1 *= 2;
        ki--;
    }
    double f = f1 - 1;
    double k = (double)ki;
    double s = f / (2 + f);
    double s2 = s * s;
    double s4 = s2 * s2;
    double t1 = s2 * (L1 + s4 * (L3 + s4 * (L5 + s4 * L7)));
    double t2 = s4 * (L2 + s4 * (L4 + s4 * L6));
    double R = t1 + t2;
    double hfsq = 0.5 * f * f;
    return k * Ln2Hi - ((hfsq - (s * (hfsq + R) + k * Ln2Lo)) - f);
}

/* Go math.log2: frac,exp := Frexp(x); if frac == 0.5 { return exp-1 }; return Log(frac)*(1/Ln2) + exp */
double lmo_go_log2(double x) {
    int e;
    double frac = frexp(x, &e);
    if (frac == 0.5) return (double)(e - 1);
    /* Go folds the untyped constant 1/Ln2 exactly before rounding to float64 */
    static const double InvLn2 = 1.44269504088896340735992468100189214;
    return go_log(frac) * InvLn2 + (double)e;
}

float lmo_seed_weight(float l) { return 0.1f * l * l; } /* lib-chaining.go:635 */

float lmo_gap_score(float gap) { /* lib-chaining.go:662 */
    if (gap == 0) return 0;
    return 0.1f * gap + 0.5f * (float)lmo_go_log2((double)gap);
}

static inline int8_t direction(const lmo_sub *a, const lmo_sub *b) { return a->tbegin >= b->tbegin ? 1 : -1; }

static inline float gap(const lmo_sub *a, const lmo_sub *b) { /* lib-chaining.go:655 */
    if (a->tbegin >= b->tbegin)
        return (float)fabs(fabs((double)(a->qbegin - b->qbegin)) - fabs((double)(a->tbegin - b->tbegin)));
    return (float)fabs(fabs((double)(a->qbegin - b->qbegin)) -
                       fabs((double)(a->tbegin + (int32_t)a->len - b->tbegin - (int32_t)b->len)));
}

static inline uint32_t f32bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
static inline float f32frombits(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}

static int cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : x > y;
}
static int cmp_i32(const void *a, const void *b) {
    int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return x < y ? -1 : x > y;
}

float lmo_chainer(const lmo_sub *subs, int n, float max_gap, float min_score, float max_distance, int top_chains,
                int **chain_off_out, int **chain_idx_out, int *nchains_out) {
    /* #chains <= n+1 and total entries <= 2n+2 (each chain marks >=1 new anchor visited, plus one extra entry
     * per chain on a direction change) */
    int *coff = (int *)malloc(sizeof(int) * (n + 4));
    int *cidx = (int *)malloc(sizeof(int) * (2 * n + 6));
    int nchains = 0, nidx = 0;
    coff[0] = 0;
    if (n == 1) { /* lib-chaining.go:125-138 */
        float w = lmo_seed_weight((float)subs[0].len);
        if (w >= min_score) {
            cidx[nidx++] = 0;
            coff[++nchains] = nidx;
        }
        *chain_off_out = coff;
        *chain_idx_out = cidx;
        *nchains_out = nchains;
        return w;
    }
    uint64_t *msi = (uint64_t *)malloc(sizeof(uint64_t) * n);
    int8_t *dirs = (int8_t *)malloc(n);
    uint64_t *s2i = (uint64_t *)malloc(sizeof(uint64_t) * n);
    uint64_t *ri = (uint64_t *)malloc(sizeof(uint64_t) * n);
    int32_t *js = (int32_t *)malloc(sizeof(int32_t) * n);
    float s = lmo_seed_weight((float)subs[0].len);
    msi[0] = (uint64_t)f32bits(s) << 32;
    dirs[0] = 0;
    s2i[0] = (uint64_t)f32bits(s) << 32;
    int32_t max_dist_i = (int32_t)max_distance;
    for (int i = 0; i < n; i++) ri[i] = ((uint64_t)(uint32_t)subs[i].tbegin << 32) | (uint32_t)i;
    qsort(ri, n, sizeof(uint64_t), cmp_u64);
    for (int i = 1; i < n; i++) {
        const lmo_sub *a = &subs[i];
        int32_t aq = a->qbegin, alen = a->len;
        float m = lmo_seed_weight((float)alen);
        int mj = i;
        int8_t mdir = 0;
        uint32_t start = a->tbegin < max_dist_i ? 0 : (uint32_t)(a->tbegin - max_dist_i);
        uint32_t right = (uint32_t)(a->tbegin + max_dist_i);
        /* rangeindex.Query :77-121 */
        int lo = 0, hi = n;
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if ((ri[mid] >> 32) < (uint64_t)start)
                lo = mid + 1;
            else
                hi = mid;
        }
        int qs = lo;
        hi = n;
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if ((ri[mid] >> 32) <= (uint64_t)right)
                lo = mid + 1;
            else
                hi = mid;
        }
        int qe = lo;
        int nj = qe - qs;
        if (nj > 0) {
            for (int t = 0; t < nj; t++) js[t] = (int32_t)(ri[qs + t] & 4294967295u);
            qsort(js, nj, sizeof(int32_t), cmp_i32);
            for (int _j = nj - 1; _j >= 0; _j--) {
                int j = js[_j];
                if (j >= i) continue;
                const lmo_sub *b = &subs[j];
                if (a->qbegin == b->qbegin || a->tbegin == b->tbegin) continue;
                if (a->qbegin - b->qbegin > max_dist_i) break;
                float g = gap(a, b);
                if (g > max_gap) continue;
                int32_t length;
                float w;
                if (aq > b->qbegin + (int32_t)b->len) {
                    length = alen;
                    w = lmo_seed_weight((float)length);
                } else if (g == 0) {
                    length = aq + alen - b->qbegin;
                    w = -lmo_seed_weight((float)b->len) + lmo_seed_weight((float)length);
                } else {
                    length = aq + alen - (b->qbegin + (int32_t)b->len);
                    w = lmo_seed_weight((float)length);
                }
                int8_t dir = direction(a, b);
                if (dirs[j] == 0 || dirs[j] == dir) {
                    s = f32frombits((uint32_t)(msi[j] >> 32)) + w - lmo_gap_score(g);
                } else {
                    s = lmo_seed_weight((float)b->len) + w - lmo_gap_score(g);
                }
                if (s >= min_score && s > m) {
                    m = s;
                    mj = j;
                    mdir = dir;
                }
            }
        }
        msi[i] = ((uint64_t)f32bits(m) << 32) | (uint32_t)mj;
        dirs[i] = mdir;
        s2i[i] = ((uint64_t)f32bits(m) << 32) | (uint32_t)i;
    }
    /* backtrack, lib-chaining.go:490-632 */
    uint8_t *visited = (uint8_t *)calloc(n, 1);
    qsort(s2i, n, sizeof(uint64_t), cmp_u64);
    int imax = n - 1;
    float max_score = 0;
    int first = 1, nchecked = 0;
    int *path = (int *)malloc(sizeof(int) * (n + 1));
    for (;;) {
        nchecked++;
        if (top_chains > 0 && nchecked > top_chains) break;
        float M = 0;
        uint32_t Mi = 0;
        while (imax >= 0) {
            M = f32frombits((uint32_t)(s2i[imax] >> 32));
            Mi = (uint32_t)s2i[imax];
            if (!visited[Mi]) {
                imax--;
                break;
            }
            imax--;
        }
        if (M < min_score) break;
        int np = 0;
        int i = (int)Mi;
        if (first) {
            max_score = M;
            first = 0;
        }
        for (;;) {
            int j = (int)(msi[i] & 4294967295u);
            int change = (i != j && dirs[j] != 0 && dirs[i] != dirs[j]);
            if (visited[j] && !change) {
                np = 0;
                visited[i] = 1;
                break;
            }
            path[np++] = i;
            visited[i] = 1;
            if (i == j || change) {
                if (change) path[np++] = j;
                for (int t = np - 1; t >= 0; t--) cidx[nidx++] = path[t];
                coff[++nchains] = nidx;
                break;
            } else {
                i = j;
            }
        }
    }
    free(path);
    free(visited);
    free(msi);
    free(dirs);
    free(s2i);
    free(ri);
    free(js);
    *chain_off_out = coff;
    *chain_idx_out = cidx;
    *nchains_out = nchains;
    return max_score;
}

/* ---------------------------------------------------------------------------------------------
 * Chainer2 (lib-chaining2.go) */
typedef struct {
    lmo_chain2 *v;
    int n, cap;
} c2vec;
static lmo_chain2 *c2push(c2vec *p) {
    if (p->n == p->cap) {
        p->cap = p->cap ? p->cap * 2 : 8;
        p->v = (lmo_chain2 *)realloc(p->v, sizeof(lmo_chain2) * p->cap);
    }
    lmo_chain2 *c = &p->v[p->n++];
    memset(c, 0, sizeof *c);
    c->alive = 1;
    return c;
}

typedef struct {
    double score;
    int qb, qe, tb, te;
} region_ret;

/* lib-chaining2.go:360-658 */
static region_ret chain_a_region(const lmo_sub *subs, const uint64_t *msi, int n, int offset, double min_score,
                                 int min_align_len, c2vec *paths, int *tot_mb, int *tot_abq, int *tot_abt, int Mi0,
                                 double hpt) {
    region_ret ret;
    double m, M = 0;
    int i, Mi = 0;
    if (Mi0 < 0) {
        for (i = 0; i < n; i++) {
            m = (double)(msi[i] >> 32);
            if (m > M) {
                M = m;
                Mi = i;
            }
        }
        if (M < min_score) {
            ret.score = 0;
            ret.qb = ret.qe = ret.tb = ret.te = -1;
            return ret;
        }
    } else {
        Mi = Mi0;
    }
    int n_matched = 0, n_abq = 0, n_abt = 0;
    i = Mi;
    int j = 0;
    int32_t qb = 0, qe = 0, tb = 0, te = 0;
    const lmo_sub *sub;
    int begin_of_next = 0;
    double pident;
    int first_anchor = 1;
    int n_anchors = 0;
    for (;;) {
        j = (int)(msi[i] & 4294967295u) - offset;
        if (j < 0) break;
        sub = &subs[i];
        n_anchors++;
        if (first_anchor) {
            first_anchor = 0;
            qe = sub->qbegin + (int32_t)sub->len - 1;
            te = sub->tbegin + (int32_t)sub->len - 1;
            qb = sub->qbegin;
            tb = sub->tbegin;
            n_matched += sub->len;
        } else {
            qb = sub->qbegin;
            tb = sub->tbegin;
            if ((int)sub->qbegin + (int)sub->len - 1 >= begin_of_next)
                n_matched += begin_of_next - (int)sub->qbegin;
            else
                n_matched += sub->len;
        }
        begin_of_next = sub->qbegin;
        if (i == j) {
            if (first_anchor) break;
            n_abq += (int)qe - (int)qb + 1;
            if (n_abq < min_align_len) {
                first_anchor = 1;
                break;
            }
            n_abt += (int)te - (int)tb + 1;
            pident = (double)n_matched / (double)(n_abq > n_abt ? n_abq : n_abt) * 100;
            if (pident < hpt) {
                first_anchor = 1;
                break;
            }
            if (pident > 100) pident = 100;
            lmo_chain2 *p = c2push(paths);
            p->nanchors = n_anchors;
            p->aligned_bases_q = n_abq;
            p->aligned_bases_t = n_abt;
            p->matched_bases = n_matched;
            p->pident = pident;
            p->qbegin = qb;
            p->qend = qe;
            p->tbegin = tb;
            p->tend = te;
            *tot_abq += n_abq;
            *tot_abt += n_abt;
            *tot_mb += n_matched;
            first_anchor = 1;
            break;
        }
        i = j;
    }
    if (j < 0) {
        if (n_anchors > 0) {
            n_abq += (int)qe - (int)qb + 1;
            n_abt += (int)te - (int)tb + 1;
            if (n_abq >= min_align_len) {
                pident = (double)n_matched / (double)(n_abq > n_abt ? n_abq : n_abt) * 100;
                if (pident >= hpt) {
                    if (pident > 100) pident = 100;
                    lmo_chain2 *p = c2push(paths);
                    p->nanchors = n_anchors;
                    p->aligned_bases_q = n_abq;
                    p->aligned_bases_t = n_abt;
                    p->matched_bases = n_matched;
                    p->pident = pident;
                    p->qbegin = qb;
                    p->qend = qe;
                    p->tbegin = tb;
                    p->tend = te;
                    *tot_abq += n_abq;
                    *tot_abt += n_abt;
                    *tot_mb += n_matched;
                }
            }
        }
    }
    int qB = qb, qE = qe, tB = tb, tE = te;
    if (Mi != n - 1) {
        region_ret r = chain_a_region(subs + Mi + 1, msi + Mi + 1, n - Mi - 1, offset + Mi + 1, min_score,
                                      min_align_len, paths, tot_mb, tot_abq, tot_abt, -1, hpt);
        if (r.score > 0) {
            if (r.qb < qB) qB = r.qb;
            if (r.qe > qE) qE = r.qe;
            if (r.tb < tB) tB = r.tb;
            if (r.te > tE) tE = r.te;
        }
    }
    if (i > 0) {
        region_ret r = chain_a_region(subs, msi, i, offset, min_score, min_align_len, paths, tot_mb, tot_abq, tot_abt,
                                      -1, hpt);
        if (r.score > 0) {
            if (r.qb < qB) qB = r.qb;
            if (r.qe > qE) qE = r.qe;
            if (r.tb < tB) tB = r.tb;
            if (r.te > tE) tE = r.te;
        }
    }
    ret.score = M;
    ret.qb = qB;
    ret.qe = qE;
    ret.tb = tB;
    ret.te = tE;
    return ret;
}

int lmo_chainer2(const lmo_sub *subs, int n, const lmo_chain2_opt *opt, lmo_chain2 **out, int *aligned_q) {
    *out = NULL;
    if (aligned_q) *aligned_q = 0;
    if (n == 1) { /* lib-chaining2.go:155-180 */
        int slen = subs[0].len;
        if (slen >= opt->min_score && slen >= opt->min_align_len) {
            c2vec paths = {0};
            lmo_chain2 *p = c2push(&paths);
            p->qbegin = subs[0].qbegin;
            p->qend = subs[0].qbegin + slen - 1;
            p->tbegin = subs[0].tbegin;
            p->tend = subs[0].tbegin + slen - 1;
            p->matched_bases = slen;
            p->pident = 100;
            p->aligned_bases_q = slen;
            p->nanchors = 1;
            *out = paths.v;
            if (aligned_q) *aligned_q = slen;
            return 1;
        }
        return 0;
    }
    int32_t band_base = opt->band_base;
    int band_count = opt->band_count;
    uint64_t *msi = (uint64_t *)malloc(sizeof(uint64_t) * n);
    msi[0] = (uint64_t)subs[0].len << 32;
    double s, m, M = 0, g;
    int mj, Mi = 0;
    double max_gap = (double)opt->max_gap;
    for (int i = 1; i < n; i++) {
        const lmo_sub *a = &subs[i];
        m = (double)a->len;
        mj = i;
        int32_t aq = a->qbegin, at = a->tbegin;
        int j = i, bcount = 0;
        for (;;) {
            j--;
            if (j < 0) break;
            const lmo_sub *b = &subs[j];
            int32_t bq = b->qbegin, bt = b->tbegin;
            if (bq == aq || bt > at) continue;
            bcount++;
            int32_t bbase = aq - bq - (int32_t)b->len;
            if (!(bbase <= band_base || bcount <= band_count)) break;
            int32_t qd = aq - bq, td = at - bt;
            if (qd < 0) qd = -qd;
            if (td < 0) td = -td;
            g = qd > td ? (double)(qd - td) : (double)(td - qd);
            if (g > max_gap) continue;
            s = (double)(msi[j] >> 32) + (double)b->len - g;
            if (s >= m) {
                m = s;
                mj = j;
            }
        }
        msi[i] = ((uint64_t)m << 32) | (uint64_t)(uint32_t)mj;
        if (m > M) {
            M = m;
            Mi = i;
        }
    }
    double min_score = (double)opt->min_score;
    if (M < min_score) {
        free(msi);
        return 0;
    }
    c2vec paths = {0};
    int tmb = 0, tabq = 0, tabt = 0;
    chain_a_region(subs, msi, n, 0, min_score, opt->min_align_len, &paths, &tmb, &tabq, &tabt, Mi,
                   opt->heuristic_pident);
    free(msi);
    if (paths.n == 0) {
        free(paths.v);
        return 0;
    }
    *out = paths.v;
    if (aligned_q) *aligned_q = tabq;
    return paths.n;
}

/* ---------------------------------------------------------------------------------------------
 * Chainer3 (lib-chaining3.go), DefaultChaining3Options :39-47 */
static inline double distance2(int32_t aq, int32_t at, int32_t bq, int32_t bt) {
    double x = fabs((double)(aq - bq)), y = fabs((double)(at - bt));
    return x > y ? x : y;
}
static inline double gap2(int32_t aq, int32_t at, int32_t bq, int32_t bt) {
    return fabs(fabs((double)(aq - bq)) - fabs((double)(at - bt)));
}

int lmo_chainer3(const lmo_sub *subs, int n, int *qend_out, int *tend_out) {
    const int32_t band_base = 10;
    const int band_count = 20;
    const double max_gap = 5, max_distance = 10, min_score = 1;
    const int min_align_len = 2;
    if (n <= 0) return 0;
    int64_t *msi = (int64_t *)malloc(sizeof(int64_t) * n);
    double s, m, M = 0, g, d;
    int mj, Mi = 0;
    const lmo_sub *a = &subs[0];
    m = (double)a->len - distance2(0, 0, a->qbegin, a->tbegin) - gap2(0, 0, a->qbegin, a->tbegin);
    msi[0] = (int64_t)((uint64_t)(int64_t)m << 32);
    for (int i = 1; i < n; i++) {
        a = &subs[i];
        m = (double)a->len - distance2(0, 0, a->qbegin, a->tbegin) - gap2(0, 0, a->qbegin, a->tbegin);
        mj = i;
        int j = i, bcount = 0;
        for (;;) {
            j--;
            if (j < 0) break;
            const lmo_sub *b = &subs[j];
            if (b->qbegin == a->qbegin || b->tbegin > a->tbegin) continue;
            bcount++;
            int32_t bbase = a->qbegin - b->qbegin - (int32_t)b->len;
            if (!(bbase <= band_base || bcount <= band_count)) break;
            d = distance2(a->qbegin, a->tbegin, b->qbegin, b->tbegin);
            if (d > max_distance) continue;
            g = gap2(a->qbegin, a->tbegin, b->qbegin, b->tbegin);
            if (g > max_gap) continue;
            s = (double)(msi[j] >> 32) + (double)b->len - d - g;
            if (s >= m) {
                m = s;
                mj = j;
            }
        }
        msi[i] = (int64_t)(((uint64_t)(int64_t)m << 32) | (uint64_t)(int64_t)mj);
        if (m > M) {
            M = m;
            Mi = i;
        }
    }
    if (M < min_score) {
        free(msi);
        return 0;
    }
    int n_matched = 0, n_abq = 0, n_abt = 0;
    int i = Mi, j;
    int32_t qb = 0, qe = 0, tb = 0, te = 0;
    int begin_of_next = 0, first_anchor = 1;
    double pident;
    int found = 0;
    for (;;) {
        j = (int)(msi[i] & 4294967295ll);
        if (j < 0) break;
        const lmo_sub *sub = &subs[i];
        if (first_anchor) {
            first_anchor = 0;
            qe = sub->qbegin + (int32_t)sub->len - 1;
            te = sub->tbegin + (int32_t)sub->len - 1;
            qb = sub->qbegin;
            tb = sub->tbegin;
            n_matched += sub->len;
        } else {
            qb = sub->qbegin;
            tb = sub->tbegin;
            if ((int)sub->qbegin + (int)sub->len - 1 >= begin_of_next)
                n_matched += begin_of_next - (int)sub->qbegin;
            else
                n_matched += sub->len;
        }
        begin_of_next = sub->qbegin;
        if (i == j) {
            n_abq += (int)qe - (int)qb + 1;
            if (n_abq < min_align_len) break;
            n_abt += (int)te - (int)tb + 1;
            pident = (double)n_matched / (double)(n_abq > n_abt ? n_abq : n_abt) * 100;
            if (pident < 15) break;
            *qend_out = qe;
            *tend_out = te;
            found = 1;
            break;
        }
        i = j;
    }
    free(msi);
    return found;
}

/* ---------------------------------------------------------------------------------------------
 * TrimSubStrPairs (lib-seq_compare.go:553-634) */
static inline float distance_f32(const lmo_sub *a, const lmo_sub *b) { /* lib-chaining.go:639 */
    double x = fabs((double)(a->qbegin - b->qbegin)), y = fabs((double)(a->tbegin - b->tbegin));
    return (float)(x > y ? x : y);
}
static inline int32_t overlap(const lmo_sub *a, const lmo_sub *b) {
    int32_t qo = 0, to = 0;
    if (b->qbegin >= a->qbegin && b->qbegin <= a->qbegin + (int32_t)a->len)
        qo = a->qbegin + (int32_t)a->len - b->qbegin + 1;
    if (b->tbegin >= a->tbegin && b->tbegin <= a->tbegin + (int32_t)a->len)
        to = a->tbegin + (int32_t)a->len - b->tbegin + 1;
    return qo > to ? qo : to;
}

int lmo_trim_subs(lmo_sub *subs, int n, int k, float min_dist, int *start_out) {
    (void)k;
    if (start_out) *start_out = 0;
    if (n < 2) return n;
    int last = n - 1;
    const lmo_sub *_p = &subs[0];
    int start = 0;
    for (int i = 0; i < n - 1; i++) { /* i indexes (*subs)[1:] */
        const lmo_sub *p = &subs[i + 1];
        if (distance_f32(p, _p) < min_dist &&
            ((p->qbegin == _p->qbegin || p->tbegin == _p->tbegin) ||
             (gap2(_p->qbegin, _p->tbegin, p->qbegin, p->tbegin) > 11 &&
              (double)overlap(_p, p) / (double)_p->len > 0.8))) {
            start = i;
            _p = p;
            continue;
        }
        break;
    }
    _p = &subs[last];
    int end = last;
    for (int i = n - 2; i >= 0; i--) {
        const lmo_sub *p = &subs[i];
        if (distance_f32(p, _p) < min_dist &&
            ((p->qbegin == _p->qbegin || p->tbegin == _p->tbegin) ||
             (gap2(p->qbegin, p->tbegin, _p->qbegin, _p->tbegin) > 11 &&
              (double)overlap(p, _p) / (double)_p->len > 0.8))) {
            end = i;
            _p = p;
            continue;
        }
        break;
    }
    if (start >= end) return 0;
    if (start > 0) memmove(subs, subs + start, sizeof(lmo_sub) * (end - start + 1));
    if (start_out) *start_out = start;
    return end - start + 1;
}
