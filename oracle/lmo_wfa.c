/* CPU ORACLE (test infrastructure only) — gap-affine wavefront alignment (WFA), end-to-end.
 *
 * The reference delegates this to github.com/shenwei356/wfa v0.5.0 (go.mod:26), which is NOT in /root/reference.
 * Restated from the published WFA algorithm (Marco-Sola et al. 2021) with WFA2-lib's semantics, which the reference's
 * changelog names as the behaviour of that module ("standard end-to-end global alignment semantics and WFA2-compatible
 * tie-breaking", CHANGELOG.md:27-29):
 *   call sites  wfa.New(DefaultPenalties,{GlobalAlignment:true}) + AdaptiveReduction(DefaultAdaptiveOption)
 *               lib-index-search.go:1842,1910-1915 ; Align(q,t) :2261,2528 ; result use :2267-2302,2327-2349
 *   penalties   mismatch 4, gap-open 6, gap-extend 2 ; wf-adaptive (min wavefront length 10, max distance diff 50,
 *               applied every step)
 *   operand order Align(query,target): pattern=query (v), text=target (h), diagonal k=h-v, offset=h.
 *               'I' consumes target only, 'D' consumes query only (the reference swaps them when printing SAM CIGAR,
 *               lib-index-search.go:2332-2339).
 *   backtrace priority on equal offsets: mismatch > D-extend > D-open > I-extend > I-open (WFA2 piggy-back codes).
 *   reads outside a stored wavefront's [lo,hi] (incl. after the adaptive cut-off) are NULL.
 * Result fields mirror wfa.AlignmentResult as used by the reference: Ops (op<<32|n), 1-based QBegin/QEnd/TBegin/TEnd
 * of the region between the first and last 'M' run, and AlignLen/Matches/Gaps/GapRegions over that region.
 * PARITY UNPINNED beyond the mismatch-only CIGAR rows of demo/q.gene.fasta.lexicmap_top-2-genomes_all.tsv.
 */
#include "lmo.h"
#include <stdlib.h>
#include <string.h>
#include <limits.h>

#define WF_NULL (INT32_MIN / 2)
#define PEN_X 4
#define PEN_O 6
#define PEN_E 2

typedef struct {
    int lo, hi;   /* valid range; null iff lo > hi */
    int alo;      /* allocation base */
    int32_t *off; /* off[k-alo] */
} wf_t;

static inline int wf_isnull(const wf_t *w) { return w == NULL || w->lo > w->hi; }
static inline int32_t wf_get(const wf_t *w, int k) {
    if (w == NULL || k < w->lo || k > w->hi) return WF_NULL;
    return w->off[k - w->alo];
}

typedef struct {
    wf_t *M, *I, *D;
    int n, cap;
} wfset;

static void wfset_grow(wfset *s, int need) {
    if (need <= s->cap) return;
    int nc = s->cap ? s->cap * 2 : 256;
    while (nc < need) nc *= 2;
    s->M = (wf_t *)realloc(s->M, sizeof(wf_t) * nc);
    s->I = (wf_t *)realloc(s->I, sizeof(wf_t) * nc);
    s->D = (wf_t *)realloc(s->D, sizeof(wf_t) * nc);
    for (int i = s->cap; i < nc; i++) {
        s->M[i].lo = s->I[i].lo = s->D[i].lo = 1;
        s->M[i].hi = s->I[i].hi = s->D[i].hi = -1;
        s->M[i].off = s->I[i].off = s->D[i].off = NULL;
        s->M[i].alo = s->I[i].alo = s->D[i].alo = 0;
    }
    s->cap = nc;
}

static void wf_alloc(wf_t *w, int lo, int hi) {
    w->lo = lo;
    w->hi = hi;
    w->alo = lo;
    w->off = (int32_t *)malloc(sizeof(int32_t) * (hi - lo + 1));
}

static void wf_trim(wf_t *w, int plen, int tlen) {
    int k;
    for (k = w->hi; k >= w->lo; --k) {
        int32_t off = w->off[k - w->alo];
        uint32_t h = (uint32_t)off, v = (uint32_t)(off - k);
        if (h <= (uint32_t)tlen && v <= (uint32_t)plen) break;
    }
    w->hi = k;
    for (k = w->lo; k <= w->hi; ++k) {
        int32_t off = w->off[k - w->alo];
        uint32_t h = (uint32_t)off, v = (uint32_t)(off - k);
        if (h <= (uint32_t)tlen && v <= (uint32_t)plen) break;
    }
    w->lo = k;
}

static inline const wf_t *wf_at(const wf_t *arr, int s) { return s < 0 ? NULL : &arr[s]; }

static void compute(wfset *W, int s, int plen, int tlen) {
    wfset_grow(W, s + 1);
    const wf_t *mm = wf_at(W->M, s - PEN_X), *mo = wf_at(W->M, s - PEN_O - PEN_E), *ie = wf_at(W->I, s - PEN_E),
               *de = wf_at(W->D, s - PEN_E);
    if (wf_isnull(mm)) mm = NULL;
    if (wf_isnull(mo)) mo = NULL;
    if (wf_isnull(ie)) ie = NULL;
    if (wf_isnull(de)) de = NULL;
    if (!mm && !mo && !ie && !de) return; /* outputs stay null */
    int lo = INT_MAX, hi = INT_MIN;
    if (mm) {
        if (mm->lo < lo) lo = mm->lo;
        if (mm->hi > hi) hi = mm->hi;
    }
    if (mo) {
        if (mo->lo - 1 < lo) lo = mo->lo - 1;
        if (mo->hi + 1 > hi) hi = mo->hi + 1;
    }
    if (ie) {
        if (ie->lo + 1 < lo) lo = ie->lo + 1;
        if (ie->hi + 1 > hi) hi = ie->hi + 1;
    }
    if (de) {
        if (de->lo - 1 < lo) lo = de->lo - 1;
        if (de->hi - 1 > hi) hi = de->hi - 1;
    }
    if (lo > hi) return;
    wf_t *om = &W->M[s], *oi = &W->I[s], *od = &W->D[s];
    wf_alloc(om, lo, hi);
    wf_alloc(oi, lo, hi);
    wf_alloc(od, lo, hi);
    for (int k = lo; k <= hi; k++) {
        int32_t a = wf_get(mo, k - 1), b = wf_get(ie, k - 1);
        int32_t ins = (a > b ? a : b) + 1;
        a = wf_get(mo, k + 1);
        b = wf_get(de, k + 1);
        int32_t del = a > b ? a : b;
        int32_t mis = wf_get(mm, k) + 1;
        int32_t mx = mis > ins ? mis : ins;
        if (del > mx) mx = del;
        uint32_t h = (uint32_t)mx, v = (uint32_t)(mx - k);
        if (h > (uint32_t)tlen) mx = WF_NULL;
        if (v > (uint32_t)plen) mx = WF_NULL;
        oi->off[k - lo] = ins;
        od->off[k - lo] = del;
        om->off[k - lo] = mx;
    }
    wf_trim(om, plen, tlen);
    wf_trim(oi, plen, tlen);
    wf_trim(od, plen, tlen);
}

static void equate(wf_t *dst, const wf_t *src) {
    if (dst->lo > dst->hi) return;
    if (src->lo > dst->lo) dst->lo = src->lo;
    if (src->hi < dst->hi) dst->hi = src->hi;
}

/* wf-adaptive: drop diagonals at either end that lag more than max_dist_diff behind the best diagonal */
static void cutoff(wfset *W, int s, int plen, int tlen, int min_wf_len, int max_dist_diff) {
    wf_t *m = &W->M[s];
    if (m->lo > m->hi) return;
    if (m->hi - m->lo + 1 >= min_wf_len) {
        int lo = m->lo, hi = m->hi;
        int *dist = (int *)malloc(sizeof(int) * (hi - lo + 1));
        int min_d = INT_MAX;
        for (int k = lo; k <= hi; k++) {
            int32_t off = m->off[k - m->alo];
            int d;
            if (off < 0) {
                d = -(WF_NULL);
            } else {
                int lv = plen - (off - k), lh = tlen - off;
                d = lv > lh ? lv : lh;
            }
            dist[k - lo] = d;
            if (d < min_d) min_d = d;
        }
        int ak = tlen - plen;
        int top = ak < hi ? ak : hi;
        for (int k = lo; k < top; ++k) {
            if (dist[k - lo] - min_d <= max_dist_diff) break;
            ++m->lo;
        }
        int bottom = ak > m->lo ? ak : m->lo;
        for (int k = hi; k > bottom; --k) {
            if (dist[k - lo] - min_d <= max_dist_diff) break;
            --m->hi;
        }
        free(dist);
    }
    equate(&W->I[s], m);
    equate(&W->D[s], m);
}

typedef struct {
    char *ops;
    int n, cap;
} opbuf;
static void op_push(opbuf *b, char op, int cnt) {
    if (cnt <= 0) return;
    if (b->n + cnt > b->cap) {
        while (b->n + cnt > b->cap) b->cap = b->cap ? b->cap * 2 : 1024;
        b->ops = (char *)realloc(b->ops, b->cap);
    }
    memset(b->ops + b->n, op, cnt);
    b->n += cnt;
}

#define BT_M 9
#define BT_D_EXT 4
#define BT_D_OPEN 3
#define BT_I_EXT 2
#define BT_I_OPEN 1
static inline int64_t piggy(int64_t off, int type) { return off < 0 ? (int64_t)WF_NULL * 16 : ((off << 4) | type); }

void lmo_wfa_result_free(lmo_wfa_result *r) {
    free(r->ops);
    memset(r, 0, sizeof *r);
}

int lmo_wfa_align(const uint8_t *q, int plen, const uint8_t *t, int tlen, int adaptive, lmo_wfa_result *res) {
    memset(res, 0, sizeof *res);
    wfset W;
    memset(&W, 0, sizeof W);
    wfset_grow(&W, 64);
    wf_alloc(&W.M[0], 0, 0);
    W.M[0].off[0] = 0;
    int s = 0;
    int ak = tlen - plen;
    int64_t max_s = (int64_t)8 * ((int64_t)plen + tlen) + 64;
    for (;;) {
        wf_t *m = &W.M[s];
        if (m->lo <= m->hi) {
            for (int k = m->lo; k <= m->hi; k++) {
                int32_t off = m->off[k - m->alo];
                if (off < 0) continue;
                int v = off - k, h = off;
                while (v < plen && h < tlen && q[v] == t[h]) {
                    v++;
                    h++;
                }
                m->off[k - m->alo] = h;
            }
            if (m->lo <= ak && ak <= m->hi && m->off[ak - m->alo] >= tlen) break;
            if (adaptive) cutoff(&W, s, plen, tlen, 10, 50);
        }
        s++;
        if (s > max_s) {
            for (int i = 0; i < W.cap; i++) {
                free(W.M[i].off);
                free(W.I[i].off);
                free(W.D[i].off);
            }
            free(W.M);
            free(W.I);
            free(W.D);
            return -1;
        }
        compute(&W, s, plen, tlen);
    }
    res->score = s;
    /* ---- backtrace (WFA2 wavefront_backtrace_affine) ---- */
    opbuf ob = {0};
    int score = s, k = ak;
    int32_t offset = tlen;
    int v = offset - k, h = offset;
    int matrix = 0; /* 0=M 1=I 2=D */
    while (v > 0 && h > 0 && score > 0) {
        int s_mis = score - PEN_X, s_open = score - PEN_O - PEN_E, s_ext = score - PEN_E;
        int64_t mx;
        int64_t c_mis = (int64_t)WF_NULL * 16, c_io = c_mis, c_ie = c_mis, c_do = c_mis, c_de = c_mis;
        if (matrix == 0) {
            if (s_mis >= 0) c_mis = piggy((int64_t)wf_get(&W.M[s_mis], k) + 1, BT_M);
            if (s_open >= 0) {
                c_io = piggy((int64_t)wf_get(&W.M[s_open], k - 1) + 1, BT_I_OPEN);
                c_do = piggy((int64_t)wf_get(&W.M[s_open], k + 1), BT_D_OPEN);
            }
            if (s_ext >= 0) {
                c_ie = piggy((int64_t)wf_get(&W.I[s_ext], k - 1) + 1, BT_I_EXT);
                c_de = piggy((int64_t)wf_get(&W.D[s_ext], k + 1), BT_D_EXT);
            }
        } else if (matrix == 1) {
            if (s_open >= 0) c_io = piggy((int64_t)wf_get(&W.M[s_open], k - 1) + 1, BT_I_OPEN);
            if (s_ext >= 0) c_ie = piggy((int64_t)wf_get(&W.I[s_ext], k - 1) + 1, BT_I_EXT);
        } else {
            if (s_open >= 0) c_do = piggy((int64_t)wf_get(&W.M[s_open], k + 1), BT_D_OPEN);
            if (s_ext >= 0) c_de = piggy((int64_t)wf_get(&W.D[s_ext], k + 1), BT_D_EXT);
        }
        mx = c_mis;
        if (c_io > mx) mx = c_io;
        if (c_ie > mx) mx = c_ie;
        if (c_do > mx) mx = c_do;
        if (c_de > mx) mx = c_de;
        if (mx < 0) break;
        if (matrix == 0) {
            int32_t max_off = (int32_t)(mx >> 4);
            op_push(&ob, 'M', offset - max_off);
            offset = max_off;
            v = offset - k;
            h = offset;
            if (v <= 0 || h <= 0) break;
        }
        int bt = (int)(mx & 15);
        switch (bt) {
        case BT_M:
            score = s_mis;
            matrix = 0;
            op_push(&ob, 'X', 1);
            --offset;
            break;
        case BT_I_OPEN:
            score = s_open;
            matrix = 0;
            op_push(&ob, 'I', 1);
            --k;
            --offset;
            break;
        case BT_I_EXT:
            score = s_ext;
            matrix = 1;
            op_push(&ob, 'I', 1);
            --k;
            --offset;
            break;
        case BT_D_OPEN:
            score = s_open;
            matrix = 0;
            op_push(&ob, 'D', 1);
            ++k;
            break;
        case BT_D_EXT:
            score = s_ext;
            matrix = 2;
            op_push(&ob, 'D', 1);
            ++k;
            break;
        }
        v = offset - k;
        h = offset;
    }
    if (v > 0 && h > 0) {
        int nm = v < h ? v : h;
        op_push(&ob, 'M', nm);
        v -= nm;
        h -= nm;
    }
    while (v > 0) {
        op_push(&ob, 'D', 1);
        --v;
    }
    while (h > 0) {
        op_push(&ob, 'I', 1);
        --h;
    }
    for (int i = 0; i < W.cap; i++) {
        free(W.M[i].off);
        free(W.I[i].off);
        free(W.D[i].off);
    }
    free(W.M);
    free(W.I);
    free(W.D);
    /* reverse and run-length encode */
    int nruns = 0;
    for (int i = ob.n - 1; i >= 0; i--)
        if (i == ob.n - 1 || ob.ops[i] != ob.ops[i + 1]) nruns++;
    res->ops = (uint64_t *)malloc(sizeof(uint64_t) * (nruns ? nruns : 1));
    int r = 0;
    for (int i = ob.n - 1; i >= 0;) {
        char op = ob.ops[i];
        int cnt = 0;
        while (i >= 0 && ob.ops[i] == op) {
            cnt++;
            i--;
        }
        res->ops[r++] = ((uint64_t)(uint8_t)op << 32) | (uint32_t)cnt;
    }
    res->nops = r;
    free(ob.ops);
    /* region between first and last M */
    int first = -1, last = -1;
    for (int i = 0; i < r; i++)
        if ((res->ops[i] >> 32) == 'M') {
            if (first < 0) first = i;
            last = i;
        }
    if (first >= 0) {
        int qpos = 0, tpos = 0; /* bases consumed so far */
        for (int i = 0; i < r; i++) {
            char op = (char)(res->ops[i] >> 32);
            int n = (int)(res->ops[i] & 0xffffffffu);
            if (i == first) {
                res->qbegin = qpos + 1;
                res->tbegin = tpos + 1;
            }
            if (op == 'M' || op == 'X') {
                qpos += n;
                tpos += n;
            } else if (op == 'I') {
                tpos += n;
            } else if (op == 'D') {
                qpos += n;
            }
            if (i >= first && i <= last) {
                res->align_len += (uint32_t)n;
                if (op == 'M') res->matches += (uint32_t)n;
                if (op == 'I' || op == 'D') {
                    res->gaps += (uint32_t)n;
                    res->gap_regions++;
                }
            }
            if (i == last) {
                res->qend = qpos;
                res->tend = tpos;
            }
        }
    }
    return 0;
}

/* lib-index-search-util.go:239-304 (trimOps + scoreAndEvalue(2,-3,5,2,totalBases,0.625,0.41)) */
#include <math.h>
#include <float.h>
void lmo_score_evalue(const lmo_wfa_result *r, int qlen, int64_t total_bases, int *score_out, int *bitscore_out,
                      double *evalue_out) {
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
    if (start < 0) {
        *score_out = 0;
        *bitscore_out = 0;
        *evalue_out = DBL_MAX;
        return;
    }
    int score = 0;
    for (int i = start; i <= end; i++) {
        int n = (int)(r->ops[i] & 4294967295u);
        switch ((char)(r->ops[i] >> 32)) {
        case 'M': score += n * 2; break;
        case 'X': score += n * -3; break;
        case 'I':
        case 'D':
        case 'H': score -= 5 + n * 2; break;
        }
    }
    int _score = score;
    if ((_score & 1) == 1) _score--;
    double lnK = log(0.41);
    double bit = (0.625 * (double)_score - lnK) / 0.693147180559945309417232121458176568;
    double evalue = (double)total_bases * pow(2, -bit) * (double)qlen;
    *score_out = score;
    *bitscore_out = (int)bit;
    *evalue_out = evalue;
}
