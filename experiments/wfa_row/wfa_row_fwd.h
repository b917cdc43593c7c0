// wfa_row_fwd.h - forward pass of the gap-affine WFA (x=4, o=6, e=2, wf-adaptive(10,50)) for FOUR alignments per wavefront:
// each alignment owns one row of 16 lanes, every lane NCR consecutive diagonals (W = 16*NCR per alignment).
//
// Why: k_wfa_lean (lexicmap_amd/csrc/lm_kernels.hip) gives an alignment a whole wavefront and keeps its wavefront ranges in
// scalar registers - ~300 scalar instructions per score step, and a CU has ONE scalar unit for all its wavefronts: at C2
// the kernel runs at 83 % of the chip's scalar issue rate and 41 % of the vector rate (profiles/r03_c2_pmc_sq.json).  Here the
// ranges live in vector registers (identical in the 16 lanes of a row), so the bookkeeping is vector work shared by four
// alignments, the scalar unit only runs the loop control, and the ring of the last five M / two I / two D wavefronts is in
// REGISTERS: with consecutive diagonals per lane the k-1 / k+1 neighbours are the lane's own registers except at the two
// ends (four cross-lane moves per score step), so the score loop touches LDS only for the packed sequences.
//
// Same recurrence, tie rules, trimming, cut-off and backtrace-byte format as k_wfa_lean: the rows written here are read by
// the same bt_walk / bt_replay.  One source for the device (hipcc) and for the host emulator (simt_emu.h): the WR_* macros
// are the only difference, and every cross-lane operation sits in wave-uniform control flow.
#pragma once
#include <stdint.h>

#ifndef WR_EXT_MODE
#define WR_EXT_MODE 2
#endif
#ifndef WR_NULL_OFF
#define WR_NULL_OFF (-1073741824) /* = LM_NULL_OFF */
#endif

struct WrRow {         // one alignment; every field identical in the 16 lanes of its row
    const uint8_t *q, *t;
    int32_t plen, tlen;
    int32_t *hdr2;     // {first diagonal, row offset} per even score (max_score + 4 entries)
    uint8_t *bt;       // backtrace bytes
    int32_t arena_cap; // bytes usable at bt
    int32_t max_score;
    uint32_t *qbuf, *tbuf; // LDS: the 2-bit packed sequences, seq_words + 2 words each
    int32_t valid;         // 0: no problem in this row
};
struct WrRes {
    int32_t status; // 0 aligned, 1 scratch / score overflow, 3 wider than W-2 diagonals or not plain ACGT, 4 no problem
    int32_t score;  // final score (status 0) or the width that did not fit (status 3)
    int32_t used;   // backtrace bytes written
};

WR_DEV uint32_t wr_pack_base(uint32_t c, bool *bad) {
    const uint32_t code = (c >> 1) & 3u;
    *bad |= c != ((0x47544341u >> (code << 3)) & 0xffu); // 'A','C','T','G' by code
    return code;
}
WR_DEV uint32_t wr_pack16(const uint8_t *s, int nb, bool *bad) {
    uint32_t w = 0;
    if (nb >= 16) {
        uint32_t b[4];
        __builtin_memcpy(b, s, 16);
#pragma unroll
        for (int j = 0; j < 16; j++) w = (w << 2) | wr_pack_base((b[j >> 2] >> ((j & 3) << 3)) & 0xffu, bad);
    } else {
        for (int j = 0; j < nb; j++) w = (w << 2) | wr_pack_base(s[j], bad);
        w <<= 2 * (16 - nb);
    }
    return w;
}
WR_DEV uint32_t wr_get16(const uint32_t *seq, int pos) {
    const int w = pos >> 4, sh = (pos & 15) << 1;
    const unsigned long long two = ((unsigned long long)seq[w] << 32) | seq[w + 1];
    return (uint32_t)((two << sh) >> 32);
}
WR_DEV int wr_match_run(const uint32_t *qb, const uint32_t *tb, int v, int h, int plen, int tlen) {
    const uint32_t d = wr_get16(qb, v) ^ wr_get16(tb, h);
    int nm = d ? (WR_CLZ(d) >> 1) : 16;
    const int rem = plen - v < tlen - h ? plen - v : tlen - h;
    nm = nm < rem ? nm : rem;
    return nm > 0 ? nm : 0;
}
WR_DEV int wr_dist(int32_t off, int k, int plen, int tlen) {
    if (off < 0) return 1073741824;
    const int lv = plen - (off - k), lh = tlen - off;
    return lv > lh ? lv : lh;
}
WR_DEV uint32_t wr_rowbits(unsigned long long ballot, int lane) { return (uint32_t)(ballot >> (lane & 48)) & 0xffffu; }

template <int NCR> WR_DEV void wfa_row4_forward(const WrRow &p, int seq_words, WrRes *res) {
    static_assert(NCR == 2 || NCR == 4 || NCR == 8, "2, 4 or 8 diagonals per lane");
    constexpr int W = 16 * NCR;
    constexpr int E_LO = 1 << 28, E_HI = -(1 << 28);
    const int lane = WR_LANE, l = lane & 15, row0 = lane & 48;
    const int plen = p.plen, tlen = p.tlen, ak = tlen - plen;
    const int qw = (plen + 15) >> 4, tw = (tlen + 15) >> 4;
    int status = p.valid ? 0 : 4;
    if (status == 0 && (qw > seq_words || tw > seq_words)) status = 3;
    bool bad = false;
    {
        const int nw = status == 0 ? (qw > tw ? qw : tw) : 0;
        for (int j = l; WR_BALLOT(j < nw) != 0ull; j += 16) {
            if (j < nw && j < qw) p.qbuf[j] = wr_pack16(p.q + 16 * j, plen - 16 * j, &bad);
            if (j < nw && j < tw) p.tbuf[j] = wr_pack16(p.t + 16 * j, tlen - 16 * j, &bad);
        }
        if (status == 0 && l == 0) {
            p.qbuf[qw] = p.qbuf[qw + 1] = 0;
            p.tbuf[tw] = p.tbuf[tw + 1] = 0;
        }
        if (wr_rowbits(WR_BALLOT(bad), lane) != 0u && status == 0) status = 3;
    }
    if (status == 0 && (p.max_score < 1 || p.arena_cap < 1)) status = 1;
    // the ring, by age in even scores: M[a] = M[s - 2a] (a = 0..4), I[a] / D[a] = score s - 2a (a = 0, 1); cells outside a
    // wavefront's range hold NULL.  Slot of diagonal k = k mod W; this lane holds slots l*NCR .. l*NCR + NCR-1.
    int32_t M[5][NCR], I[2][NCR], D[2][NCR];
#pragma unroll
    for (int c = 0; c < NCR; c++) {
#pragma unroll
        for (int a = 0; a < 5; a++) M[a][c] = WR_NULL_OFF;
        I[0][c] = I[1][c] = D[0][c] = D[1][c] = WR_NULL_OFF;
    }
    int mlo[5], mhi[5], ilo[2], ihi[2], dlo[2], dhi[2];
#pragma unroll
    for (int a = 0; a < 5; a++) {
        mlo[a] = E_LO;
        mhi[a] = E_HI;
    }
    ilo[0] = ilo[1] = dlo[0] = dlo[1] = E_LO;
    ihi[0] = ihi[1] = dhi[0] = dhi[1] = E_HI;
    mlo[0] = mhi[0] = 0;
    if (l == 0) M[0][0] = 0; // diagonal 0 = slot 0
    int s = 0, alo = 0, s_final = 0, wide_at = 0;
    int32_t used = 1; // score 0 = one cell that is never read
    bool live = status == 0;
    if (live && l == 0) {
        p.hdr2[0] = 0;
        p.hdr2[1] = 0;
        p.hdr2[2] = 0;
        p.hdr2[3] = 1;
    }
    WR_LDS_SYNC(); // the packed sequences are in LDS
    while (WR_BALLOT(live) != 0ull) {
        // ---- extension of M[s]
        const bool has = live && mlo[0] <= mhi[0];
        int kc[NCR], jc[NCR];
        bool inr[NCR];
        bool fin = false;
#if WR_EXT_MODE == 0
        // one ballot loop per cell of the lane (the form of k_wfa_lean): NCR dependent chains of LDS round trips
#pragma unroll
        for (int c = 0; c < NCR; c++) {
            const int slot = l * NCR + c;
            const int j = (slot - alo) & (W - 1);
            const int k = alo + j;
            kc[c] = k;
            jc[c] = j;
            inr[c] = has && (uint32_t)(k - mlo[0]) <= (uint32_t)(mhi[0] - mlo[0]);
            const int32_t o = M[0][c];
            const bool act = inr[c] && o >= 0;
            int v = act ? o - k : 0, h = act ? o : 0;
            bool ext = act;
            while (WR_BALLOT(ext) != 0ull) { // 16 bases per pass; lanes that are done read position 0 and add nothing
                const int run = wr_match_run(p.qbuf, p.tbuf, v, h, plen, tlen);
                const int nm = ext ? run : 0;
                v += nm;
                h += nm;
                ext = nm == 16;
            }
            if (act) M[0][c] = h;
            fin = fin || (inr[c] && k == ak && M[0][c] >= tlen);
        }
#else
        // the NCR cells of a lane side by side: their LDS reads are independent, so a pass costs one LDS round trip instead of
        // NCR, and the number of passes is the longest run of the wavefront / 16 instead of the sum over the cells
        int hx[NCR], kx[NCR]; // offset and diagonal of the cells being extended; (0, 0) for an idle cell: it reads position 0
        bool ext[NCR];
        bool any = false;
#pragma unroll
        for (int c = 0; c < NCR; c++) {
            const int slot = l * NCR + c;
            const int j = (slot - alo) & (W - 1);
            const int k = alo + j;
            kc[c] = k;
            jc[c] = j;
            inr[c] = has && (uint32_t)(k - mlo[0]) <= (uint32_t)(mhi[0] - mlo[0]);
            const int32_t o = M[0][c];
            ext[c] = inr[c] && o >= 0;
            hx[c] = ext[c] ? o : 0;
            kx[c] = ext[c] ? k : 0;
            any = any || ext[c];
        }
#if WR_EXT_MODE == 1
        while (WR_BALLOT(any) != 0ull) {
            any = false;
#pragma unroll
            for (int c = 0; c < NCR; c++) {
                const int run = wr_match_run(p.qbuf, p.tbuf, hx[c] - kx[c], hx[c], plen, tlen);
                const int nm = ext[c] ? run : 0;
                hx[c] += nm;
                ext[c] = nm == 16;
                any = any || ext[c];
            }
        }
#else
        // first pass: every cell; later passes (the few cells on a long run): one cell per lane and pass, the first still running
        if (WR_BALLOT(any) != 0ull) {
            any = false;
#pragma unroll
            for (int c = 0; c < NCR; c++) {
                const int run = wr_match_run(p.qbuf, p.tbuf, hx[c] - kx[c], hx[c], plen, tlen);
                const int nm = ext[c] ? run : 0;
                hx[c] += nm;
                ext[c] = nm == 16;
                any = any || ext[c];
            }
            while (WR_BALLOT(any) != 0ull) {
                int sel = NCR - 1;
#pragma unroll
                for (int c = NCR - 2; c >= 0; c--) sel = ext[c] ? c : sel;
                int hh = hx[NCR - 1], kq = kx[NCR - 1];
#pragma unroll
                for (int c = NCR - 2; c >= 0; c--) {
                    hh = sel == c ? hx[c] : hh;
                    kq = sel == c ? kx[c] : kq;
                }
                const int run = wr_match_run(p.qbuf, p.tbuf, hh - kq, hh, plen, tlen);
                const int nm = any ? run : 0;
                bool more = false;
#pragma unroll
                for (int c = 0; c < NCR; c++) {
                    if (sel == c) {
                        hx[c] += nm;
                        ext[c] = ext[c] && nm == 16;
                    }
                    more = more || ext[c];
                }
                any = more;
            }
        }
#endif
#pragma unroll
        for (int c = 0; c < NCR; c++) {
            if (inr[c] && M[0][c] >= 0) M[0][c] = hx[c];
            fin = fin || (inr[c] && kc[c] == ak && M[0][c] >= tlen);
        }
#endif
        const bool done = wr_rowbits(WR_BALLOT(fin), lane) != 0u;
        // ---- wf-adaptive(10, 50)
        const bool adapt = has && !done && mhi[0] - mlo[0] + 1 >= 10;
        if (WR_BALLOT(adapt) != 0ull) {
            int dist[NCR];
            int dm = 2147483647;
#pragma unroll
            for (int c = 0; c < NCR; c++) {
                dist[c] = inr[c] ? wr_dist(M[0][c], kc[c], plen, tlen) : 2147483647;
                dm = dist[c] < dm ? dist[c] : dm;
            }
            const int dmin = (int)WR_ROW_MIN_I32(dm);
            const int top = ak < mhi[0] ? ak : mhi[0];
            const int bottom = ak > mlo[0] ? ak : mlo[0];
            uint32_t enc = 0xffffffffu;
#pragma unroll
            for (int c = 0; c < NCR; c++) {
                const bool keep = inr[c] && (dist[c] - dmin <= 50);
                const uint32_t l16 = (keep && kc[c] < top) ? (uint32_t)jc[c] : 0xffffu;
                const uint32_t h16 = (keep && kc[c] > bottom) ? (uint32_t)(W - 1 - jc[c]) : 0xffffu;
                enc = wr_pk_min_u16(enc, l16 | (h16 << 16));
            }
            const uint32_t red = WR_ROW_PKMIN_U16(enc);
            int nlo = mlo[0], nhi = mhi[0];
            if (mlo[0] < top) nlo = (red & 0xffffu) != 0xffffu ? alo + (int)(red & 0xffffu) : top;
            if (mhi[0] > bottom) nhi = (red >> 16) != 0xffffu ? alo + (W - 1 - (int)(red >> 16)) : bottom;
            if (adapt && (nlo != mlo[0] || nhi != mhi[0])) {
                const int oil = ilo[0], odl = dlo[0];
                const uint32_t oisp = (uint32_t)(ihi[0] - ilo[0]), odsp = (uint32_t)(dhi[0] - dlo[0]);
                ilo[0] = ilo[0] > nlo ? ilo[0] : nlo;
                ihi[0] = ihi[0] < nhi ? ihi[0] : nhi;
                dlo[0] = dlo[0] > nlo ? dlo[0] : nlo;
                dhi[0] = dhi[0] < nhi ? dhi[0] : nhi;
                if (ilo[0] > ihi[0]) {
                    ilo[0] = E_LO;
                    ihi[0] = E_HI;
                }
                if (dlo[0] > dhi[0]) {
                    dlo[0] = E_LO;
                    dhi[0] = E_HI;
                }
                const uint32_t nmsp = (uint32_t)(nhi - nlo), nisp = (uint32_t)(ihi[0] - ilo[0]), ndsp = (uint32_t)(dhi[0] - dlo[0]);
#pragma unroll
                for (int c = 0; c < NCR; c++) {
                    const int k = kc[c];
                    if (inr[c] && (uint32_t)(k - nlo) > nmsp) M[0][c] = WR_NULL_OFF;
                    if ((uint32_t)(k - oil) <= oisp && (uint32_t)(k - ilo[0]) > nisp) I[0][c] = WR_NULL_OFF;
                    if ((uint32_t)(k - odl) <= odsp && (uint32_t)(k - dlo[0]) > ndsp) D[0][c] = WR_NULL_OFF;
                }
                mlo[0] = nlo;
                mhi[0] = nhi;
            }
        }
        if (done) {
            live = false;
            s_final = s;
        }
        // ---- next even score
        if (live) {
            s += 2;
            if (s >= p.max_score) {
                status = 1;
                live = false;
            }
        }
#pragma unroll
        for (int a = 4; a > 0; a--) {
            mlo[a] = mlo[a - 1];
            mhi[a] = mhi[a - 1];
#pragma unroll
            for (int c = 0; c < NCR; c++) M[a][c] = M[a - 1][c];
        }
        ilo[1] = ilo[0];
        ihi[1] = ihi[0];
        dlo[1] = dlo[0];
        dhi[1] = dhi[0];
#pragma unroll
        for (int c = 0; c < NCR; c++) {
            I[1][c] = I[0][c];
            D[1][c] = D[0][c];
        }
        // sources: M[s-4] (mismatch), M[s-8] (gap open), I[s-2] / D[s-2] (gap extension)
        int lo = mlo[2] < mlo[4] - 1 ? mlo[2] : mlo[4] - 1, hi = mhi[2] > mhi[4] + 1 ? mhi[2] : mhi[4] + 1;
        {
            const int l2 = ilo[1] + 1 < dlo[1] - 1 ? ilo[1] + 1 : dlo[1] - 1, h2 = ihi[1] + 1 > dhi[1] - 1 ? ihi[1] + 1 : dhi[1] - 1;
            lo = l2 < lo ? l2 : lo;
            hi = h2 > hi ? h2 : hi;
        }
        const bool empty = live && lo > hi;
        bool comp = live && lo <= hi;
        const int wd = hi - lo + 1;
        if (comp && wd > W - 2) {
            status = 3;
            wide_at = wd;
            live = false;
            comp = false;
        }
        if (comp && (int64_t)used + wd > (int64_t)p.arena_cap) {
            status = 1;
            live = false;
            comp = false;
        }
        const int32_t rowb = used;
        if (empty) { // no source wavefront: an empty row with the offset of the next one
            mlo[0] = ilo[0] = dlo[0] = E_LO;
            mhi[0] = ihi[0] = dhi[0] = E_HI;
            alo = 0;
            if (l == 0) {
                p.hdr2[s] = 0;
                p.hdr2[s + 1] = used;
                p.hdr2[s + 3] = used;
            }
        }
        if (comp) {
            used += wd;
            alo = lo;
            if (l == 0) { // entry s/2 = {lo, row offset}; the offset of entry s/2+1 closes the row
                p.hdr2[s] = lo;
                p.hdr2[s + 1] = rowb;
                p.hdr2[s + 3] = used;
            }
        }
        // the two cells beside this lane's run of diagonals: slot-1 of its first cell, slot+1 of its last (the row is a ring)
        const int lm1 = row0 | ((l + 15) & 15), lp1 = row0 | ((l + 1) & 15);
        const int32_t mL = (int32_t)WR_SHFL((uint32_t)M[4][NCR - 1], lm1), iL = (int32_t)WR_SHFL((uint32_t)I[1][NCR - 1], lm1);
        const int32_t mR = (int32_t)WR_SHFL((uint32_t)M[4][0], lp1), dR = (int32_t)WR_SHFL((uint32_t)D[1][0], lp1);
        int kk[NCR];
        int32_t vins[NCR], vdel[NCR], vmx[NCR];
        uint32_t em = 0xffffffffu, ei = 0xffffffffu, ed = 0xffffffffu; // (first, W-1-last) cell inside the DP matrix
#pragma unroll
        for (int c = 0; c < NCR; c++) {
            const int slot = l * NCR + c;
            const int j = (slot - lo) & (W - 1);
            const int k = lo + j;
            kk[c] = k;
            const bool in = comp && k <= hi;
            int32_t a = c > 0 ? M[4][c > 0 ? c - 1 : 0] : mL, b = c > 0 ? I[1][c > 0 ? c - 1 : 0] : iL;
            const bool iext = b >= a; // equal offsets: extension
            const int32_t ins = (iext ? b : a) + 1;
            a = c < NCR - 1 ? M[4][c < NCR - 1 ? c + 1 : 0] : mR;
            b = c < NCR - 1 ? D[1][c < NCR - 1 ? c + 1 : 0] : dR;
            const bool dext = b >= a;
            const int32_t del = dext ? b : a;
            const int32_t mis = M[2][c] + 1;
            int32_t mx = mis > ins ? mis : ins;
            if (del > mx) mx = del;
            // predecessor of the M cell on equal offsets: mismatch > deletion > insertion
            const uint32_t mc = (mis >= del && mis >= ins) ? 0u : (del >= ins ? 2u : 1u);
            if ((uint32_t)mx > (uint32_t)tlen) mx = WR_NULL_OFF;
            if ((uint32_t)(mx - k) > (uint32_t)plen) mx = WR_NULL_OFF;
            if (in) p.bt[rowb + (k - lo)] = (uint8_t)(mc | (iext ? 4u : 0u) | (dext ? 8u : 0u));
            vins[c] = ins;
            vdel[c] = del;
            vmx[c] = mx;
            const uint32_t pos = (uint32_t)j | ((uint32_t)(W - 1 - j) << 16);
            const bool okm = in && (uint32_t)mx <= (uint32_t)tlen && (uint32_t)(mx - k) <= (uint32_t)plen;
            const bool oki = in && (uint32_t)ins <= (uint32_t)tlen && (uint32_t)(ins - k) <= (uint32_t)plen;
            const bool okd = in && (uint32_t)del <= (uint32_t)tlen && (uint32_t)(del - k) <= (uint32_t)plen;
            em = wr_pk_min_u16(em, okm ? pos : 0xffffffffu);
            ei = wr_pk_min_u16(ei, oki ? pos : 0xffffffffu);
            ed = wr_pk_min_u16(ed, okd ? pos : 0xffffffffu);
        }
        const uint32_t rm = WR_ROW_PKMIN_U16(em), ri = WR_ROW_PKMIN_U16(ei), rd = WR_ROW_PKMIN_U16(ed);
        if (comp) {
            const bool hm = (rm & 0xffffu) != 0xffffu, hi_ = (ri & 0xffffu) != 0xffffu, hd = (rd & 0xffffu) != 0xffffu;
            mlo[0] = hm ? lo + (int)(rm & 0xffffu) : E_LO;
            mhi[0] = hm ? lo + (W - 1 - (int)(rm >> 16)) : E_HI;
            ilo[0] = hi_ ? lo + (int)(ri & 0xffffu) : E_LO;
            ihi[0] = hi_ ? lo + (W - 1 - (int)(ri >> 16)) : E_HI;
            dlo[0] = hd ? lo + (int)(rd & 0xffffu) : E_LO;
            dhi[0] = hd ? lo + (W - 1 - (int)(rd >> 16)) : E_HI;
        }
        const uint32_t spm = (uint32_t)(mhi[0] - mlo[0]), spi = (uint32_t)(ihi[0] - ilo[0]), spd = (uint32_t)(dhi[0] - dlo[0]);
#pragma unroll
        for (int c = 0; c < NCR; c++) { // (k - E_LO) as unsigned is above every span, also above the span of an empty range
            const int k = kk[c];
            M[0][c] = comp && (uint32_t)(k - mlo[0]) <= spm ? vmx[c] : WR_NULL_OFF;
            I[0][c] = comp && (uint32_t)(k - ilo[0]) <= spi ? vins[c] : WR_NULL_OFF;
            D[0][c] = comp && (uint32_t)(k - dlo[0]) <= spd ? vdel[c] : WR_NULL_OFF;
        }
    }
    res->status = status;
    res->score = status == 0 ? s_final : (status == 3 ? wide_at : 0);
    res->used = used;
}
