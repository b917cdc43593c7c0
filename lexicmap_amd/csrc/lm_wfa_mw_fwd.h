// wfa_mw_fwd.h - forward pass of the gap-affine WFA (x=4, o=6, e=2, wf-adaptive(10,50)) by a WORKGROUP of four wavefronts
// per alignment: 256 threads x NCW cells = 512 (NCW 2) or 1024 (NCW 4) diagonals.
//
// Why: the 512 / 1024-diagonal launches of k_wfa_lean (lexicmap_amd/csrc/lm_kernels.hip) are a handful of 30-50-kb alignments
// each run by ONE wavefront with 8 / 16 cells per lane: 60-90 ms per alignment at single-wavefront latency, 0.2-4 % of the
// chip's issue rate, and every round of the C3 pipeline waits for the chain 256 -> 512 -> 1024 diagonals (co-scheduled 89 +
// 126 + 158 ms of a 334-ms round, profiles/r03_c3_bench.json).  Four wavefronts with 2 / 4 cells per lane run the same score
// step in about a third of the time; what they pay is three workgroup barriers per score.
//
// Same recurrence, tie rules, trimming, cut-off, ring layout (slot of diagonal k = (k + koff) mod W in LDS, NULL outside a
// wavefront's range) and backtrace-byte format as k_wfa_lean: the rows written here are read by the same bt_walk / bt_replay.
// One source for the device (hipcc) and for the host emulator (simt_emu.h); every wave-level operation sits in control flow
// that is uniform for the wavefront, every barrier in control flow that is uniform for the workgroup.
#pragma once
#include <stdint.h>

#ifndef WR_NULL_OFF
#define WR_NULL_OFF (-1073741824) /* = LM_NULL_OFF */
#endif
#define MW_THREADS 256
#define MW_WAVES 4

struct MwProb { // one alignment; identical in every thread of the workgroup
    const uint8_t *q, *t;
    int32_t plen, tlen;
    int32_t *hdr2;     // {first diagonal, row offset} per even score (max_score + 4 entries)
    uint8_t *bt;       // backtrace bytes
    int32_t arena_cap; // bytes usable at bt
    int32_t max_score;
};
struct MwLds {
    int32_t *ring;         // 9 * W words: M rows 0-4, I rows 5-6, D rows 7-8
    uint32_t *qbuf, *tbuf; // the 2-bit packed sequences, seq_words + 2 words each (WIN: the windows, MW_WINW + 2 words each)
    int32_t *red;          // 40 words of reduction scratch
};
struct MwRes {
    int32_t status; // 0 aligned, 1 scratch / score overflow, 3 wider than W-2 diagonals or not plain ACGT
    int32_t score;  // final score (status 0) or the width that did not fit (status 3)
    int32_t used;   // backtrace bytes written
};

WR_DEV uint32_t mw_pack_base(uint32_t c, bool *bad) {
    const uint32_t code = (c >> 1) & 3u;
    *bad |= c != ((0x47544341u >> (code << 3)) & 0xffu); // 'A','C','T','G' by code
    return code;
}
WR_DEV uint32_t mw_pack16(const uint8_t *s, int nb, bool *bad) {
    uint32_t w = 0;
    if (nb >= 16) {
        uint32_t b[4];
        __builtin_memcpy(b, s, 16);
#pragma unroll
        for (int j = 0; j < 16; j++) w = (w << 2) | mw_pack_base((b[j >> 2] >> ((j & 3) << 3)) & 0xffu, bad);
    } else {
        for (int j = 0; j < nb; j++) w = (w << 2) | mw_pack_base(s[j], bad);
        w <<= 2 * (16 - nb);
    }
    return w;
}
WR_DEV uint32_t mw_get16(const uint32_t *seq, int pos) {
    const int w = pos >> 4, sh = (pos & 15) << 1;
    const unsigned long long two = ((unsigned long long)seq[w] << 32) | seq[w + 1];
    return (uint32_t)((two << sh) >> 32);
}
WR_DEV int mw_match_run(const uint32_t *qb, const uint32_t *tb, int v, int h, int plen, int tlen) {
    const uint32_t d = mw_get16(qb, v) ^ mw_get16(tb, h);
    int nm = d ? (WR_CLZ(d) >> 1) : 16;
    const int rem = plen - v < tlen - h ? plen - v : tlen - h;
    nm = nm < rem ? nm : rem;
    return nm > 0 ? nm : 0;
}
// ---- sliding 2-bit windows (the WIN form): each sequence as a circular window of MW_WINW words of 16 bases (4096 bases); word w
// lives at slot w & (MW_WINW - 1), slots 0 and 1 are mirrored behind the last one so that three consecutive words never wrap
// (the layout of k_wfa_lean's WfaWin: bt_replay<true> continues on the same buffers)
#define MW_WINW 256
WR_DEV bool mw_win_has(int w0, int pos) { return (uint32_t)((pos >> 4) - w0) < (uint32_t)(MW_WINW - 2); }
WR_DEV uint64_t mw_win_get32(const uint32_t *buf, int pos) {
    const uint32_t *p = buf + ((pos >> 4) & (MW_WINW - 1));
    const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
    const int rs = 32 - ((pos & 15) << 1); // 2..32
    const uint32_t hi = (uint32_t)((((uint64_t)d0 << 32) | d1) >> rs);
    const uint32_t lo = (uint32_t)((((uint64_t)d1 << 32) | d2) >> rs);
    return ((uint64_t)hi << 32) | lo;
}
WR_DEV int mw_match_run_win(const uint32_t *qb, const uint32_t *tb, int v, int h, int plen, int tlen) { // at most 32 bases
    const uint64_t d = mw_win_get32(qb, v) ^ mw_win_get32(tb, h);
    int nm = d ? (WR_CLZLL(d) >> 1) : 32;
    const int rem = plen - v < tlen - h ? plen - v : tlen - h;
    nm = nm < rem ? nm : rem;
    return nm > 0 ? nm : 0;
}
// the whole workgroup makes words [qw0, qw0 + WINW) of Q and [tw0, tw0 + WINW) of T resident; words that stay are not reloaded
WR_DEV void mw_win_move2(uint32_t *qbuf, const uint8_t *q, int plen, int *qw0_cur, int qw0, uint32_t *tbuf, const uint8_t *t, int tlen, int *tw0_cur,
                         int tw0, int tid, bool *bad, bool fresh) {
    WR_BARRIER(); // every thread is done reading the slots that are about to change
    const bool qkeep = !fresh && qw0 >= *qw0_cur && qw0 < *qw0_cur + MW_WINW, tkeep = !fresh && tw0 >= *tw0_cur && tw0 < *tw0_cur + MW_WINW;
    const int qfrom = qkeep ? *qw0_cur + MW_WINW : qw0, tfrom = tkeep ? *tw0_cur + MW_WINW : tw0;
    const int nq = qw0 + MW_WINW - qfrom, nt = tw0 + MW_WINW - tfrom; // words to load (0 when a window does not move)
    for (int i = tid; i < nq + nt; i += MW_THREADS) {
        const bool isq = i < nq;
        const int w = isq ? qfrom + i : tfrom + (i - nq);
        const uint8_t *src = isq ? q : t;
        uint32_t *buf = isq ? qbuf : tbuf;
        const int nb = (isq ? plen : tlen) - 16 * w;
        const uint32_t word = nb > 0 ? mw_pack16(src + 16 * (int64_t)w, nb, bad) : 0u;
        const int slot = w & (MW_WINW - 1);
        buf[slot] = word;
        if (slot < 2) buf[MW_WINW + slot] = word;
    }
    *qw0_cur = WR_UNIFORM(qw0);
    *tw0_cur = WR_UNIFORM(tw0);
    WR_BARRIER();
}

WR_DEV int mw_dist(int32_t off, int k, int plen, int tlen) {
    if (off < 0) return 1073741824;
    const int lv = plen - (off - k), lh = tlen - off;
    return lv > lh ? lv : lh;
}

// WIN: the sequences through sliding windows (L.qbuf / L.tbuf: MW_WINW + 2 words each, any length) instead of whole in LDS
template <int NCW, bool WIN> WR_DEV void wfa_mw_forward(const MwProb &p, const MwLds &L, int seq_words, MwRes *res) {
    static_assert(NCW == 1 || NCW == 2 || NCW == 4, "1, 2 or 4 diagonals per thread");
    constexpr int T = MW_THREADS, W = T * NCW;
    constexpr int E_LO = 1 << 28, E_HI = -(1 << 28);
    const int tid = WR_TID, lane = tid & 63, wave = tid >> 6;
    int32_t *const rM = L.ring, *const rI = L.ring + 5 * W, *const rD = L.ring + 7 * W; // row r of X at X + r * W
    const int plen = p.plen, tlen = p.tlen, ak = tlen - plen;
    // slot of diagonal k = (k + koff) mod W: the band between diagonal 0 and the final diagonal is centred on the ring
    const int koff = W / 2 - (ak >= -(W / 2) && ak <= W / 2 ? ak / 2 : 0);
    int status = 0;
    bool bad = false;
    int qw0 = 0, tw0 = 0; // WIN: first resident word of either window (uniform)
    if (WIN) {
        mw_win_move2(L.qbuf, p.q, plen, &qw0, 0, L.tbuf, p.t, tlen, &tw0, 0, tid, &bad, true);
    } else {
        const int qw = (plen + 15) >> 4, tw = (tlen + 15) >> 4;
        if (qw > seq_words || tw > seq_words) {
            status = 3;
        } else {
            for (int j = tid; j < qw; j += T) L.qbuf[j] = mw_pack16(p.q + 16 * j, plen - 16 * j, &bad);
            for (int j = tid; j < tw; j += T) L.tbuf[j] = mw_pack16(p.t + 16 * j, tlen - 16 * j, &bad);
            if (tid == 0) {
                L.qbuf[qw] = L.qbuf[qw + 1] = 0;
                L.tbuf[tw] = L.tbuf[tw + 1] = 0;
            }
        }
    }
    {
        const bool wbad = WR_BALLOT(bad) != 0ull;
        if (lane == 0) L.red[wave] = wbad ? 1 : 0;
    }
#pragma unroll
    for (int c = 0; c < NCW; c++) {
#pragma unroll
        for (int r = 0; r < 9; r++) L.ring[r * W + tid + T * c] = WR_NULL_OFF;
    }
    WR_BARRIER(); // sequences, ring and the flags are in LDS
    if (!WIN && status == 0 && (L.red[0] | L.red[1] | L.red[2] | L.red[3]) != 0) status = 3; // (WIN: checked when the alignment ends)
    status = WR_UNIFORM(status);
    if (status == 0 && (p.max_score < 1 || p.arena_cap < 1)) status = 1;
    int mlo[5], mhi[5], ilo[2], ihi[2], dlo[2], dhi[2];
#pragma unroll
    for (int a = 0; a < 5; a++) {
        mlo[a] = E_LO;
        mhi[a] = E_HI;
    }
    ilo[0] = ilo[1] = dlo[0] = dlo[1] = E_LO;
    ihi[0] = ihi[1] = dhi[0] = dhi[1] = E_HI;
    mlo[0] = mhi[0] = 0;
    if (tid == 0) rM[koff & (W - 1)] = 0; // row 0
    int s = 0, ms = 0, is = 0, alo = 0, wide_at = 0;
    int32_t used = 1; // score 0 = one cell that is never read
    if (tid == 0 && status == 0) {
        p.hdr2[0] = 0;
        p.hdr2[1] = 0;
        p.hdr2[2] = 0;
        p.hdr2[3] = 1;
    }
    // which of the NCW chunks of 256 slots does the slot range of diagonals [lo, hi] touch (bit c) ?
    auto chunk_mask = [&](int lo_, int hi_) -> uint32_t {
        if (NCW == 1) return 1u;
        const int s0 = (lo_ + koff) & (W - 1), s1 = s0 + (hi_ - lo_);
        const int cf = s0 / T, cl = (s1 / T) < NCW - 1 ? (s1 / T) : NCW - 1;
        uint32_t m = ((2u << (cl - cf)) - 1u) << cf;
        if (s1 >= W) m |= (2u << ((s1 - W) / T)) - 1u;
        return (uint32_t)WR_UNIFORM((int)m);
    };
    WR_BARRIER(); // rM[0][slot of 0] = 0 is visible; L.red may be written again
    while (status == 0) {
        bool done = false;
        if (mlo[0] <= mhi[0]) {
            const uint32_t cmx = chunk_mask(mlo[0], mhi[0]);
            int kc[NCW], jc[NCW];
            bool inr[NCW];
            int32_t off[NCW];
            uint32_t extm = 0; // WIN: bit c = this thread's cell of chunk c is still being extended
#pragma unroll
            for (int c = 0; c < NCW; c++) {
                const int slot = tid + T * c;
                const int j = (slot - koff - alo) & (W - 1);
                const int k = alo + j;
                kc[c] = k;
                jc[c] = j;
                inr[c] = false;
                off[c] = WR_NULL_OFF;
                if (((cmx >> c) & 1u) == 0) continue; // (uniform for the workgroup)
                inr[c] = (uint32_t)(k - mlo[0]) <= (uint32_t)(mhi[0] - mlo[0]);
                off[c] = rM[ms * W + slot];
                if (inr[c] && off[c] >= 0) extm |= 1u << c;
            }
            int dist[NCW];
            while (true) { // (one pass unless a cell has to wait for the windows)
#pragma unroll
                for (int c = 0; c < NCW; c++) {
                    if (((cmx >> c) & 1u) == 0) continue;
                    bool ext = ((extm >> c) & 1u) != 0;
                    int h = ext ? off[c] : 0, v = ext ? h - kc[c] : 0;
                    if (!WIN) {
                        while (WR_BALLOT(ext) != 0ull) { // per wavefront: 16 bases per pass from the whole packed sequences
                            const int run = mw_match_run(L.qbuf, L.tbuf, v, h, plen, tlen);
                            const int nm = ext ? run : 0;
                            v += nm;
                            h += nm;
                            ext = nm == 16;
                        }
                    } else {
                        while (true) { // 32 bases per pass from the windows; a cell outside a window waits
                            const bool in = mw_win_has(qw0, v) && mw_win_has(tw0, h);
                            if (WR_BALLOT(ext && in) == 0ull) break;
                            const int run = mw_match_run_win(L.qbuf, L.tbuf, v, h, plen, tlen);
                            const int nm = (ext && in) ? run : 0;
                            v += nm;
                            h += nm;
                            ext = ext && (!in || nm == 32);
                        }
                    }
                    if ((extm >> c) & 1u) off[c] = h;
                    if (!ext) extm &= ~(1u << c);
                }
                // across the four wavefronts: has the final cell been reached, the smallest distance to the end, and (WIN) is
                // anybody waiting for the windows - then both move to the smallest waiting positions and the extension goes on
                bool fin = false;
                int dm = 2147483647, mv = 2147483647, mh = 2147483647;
#pragma unroll
                for (int c = 0; c < NCW; c++) {
                    dist[c] = inr[c] ? mw_dist(off[c], kc[c], plen, tlen) : 2147483647;
                    dm = dist[c] < dm ? dist[c] : dm;
                    fin = fin || (inr[c] && kc[c] == ak && off[c] >= tlen);
                    const bool wt = ((extm >> c) & 1u) != 0;
                    mh = wt && off[c] < mh ? off[c] : mh;
                    mv = wt && off[c] - kc[c] < mv ? off[c] - kc[c] : mv;
                }
                {
                    const bool wfin = WR_BALLOT(fin) != 0ull;
                    const int wdm = (int)WR_WAVE_MIN_I32(dm);
                    int wmv = 0, wmh = 0;
                    bool wwait = false;
                    if (WIN) {
                        wwait = WR_BALLOT(extm != 0u) != 0ull;
                        wmv = (int)WR_WAVE_MIN_I32(mv);
                        wmh = (int)WR_WAVE_MIN_I32(mh);
                    }
                    if (lane == 0) {
                        L.red[wave] = wfin ? 1 : 0;
                        L.red[4 + wave] = wdm;
                        if (WIN) {
                            L.red[24 + wave] = wwait ? 1 : 0;
                            L.red[28 + wave] = wmv;
                            L.red[32 + wave] = wmh;
                        }
                    }
                }
#pragma unroll
                for (int c = 0; c < NCW; c++)
                    if (inr[c] && off[c] >= 0) rM[ms * W + tid + T * c] = off[c];
                WR_BARRIER(); // B1: the partial results, and every extended cell, are in LDS
                if (!WIN) break;
                if (WR_UNIFORM(L.red[24] | L.red[25] | L.red[26] | L.red[27]) == 0) break;
                int gv = L.red[28] < L.red[29] ? L.red[28] : L.red[29], gh = L.red[32] < L.red[33] ? L.red[32] : L.red[33];
                gv = L.red[30] < gv ? L.red[30] : gv;
                gv = L.red[31] < gv ? L.red[31] : gv;
                gh = L.red[34] < gh ? L.red[34] : gh;
                gh = L.red[35] < gh ? L.red[35] : gh;
                mw_win_move2(L.qbuf, p.q, plen, &qw0, WR_UNIFORM(gv) >> 4, L.tbuf, p.t, tlen, &tw0, WR_UNIFORM(gh) >> 4, tid, &bad, false);
            }
            done = WR_UNIFORM(L.red[0] | L.red[1] | L.red[2] | L.red[3]) != 0;
            if (!done && mhi[0] - mlo[0] + 1 >= 10) { // wf-adaptive(10, 50)
                int dmin = L.red[4] < L.red[5] ? L.red[4] : L.red[5];
                dmin = L.red[6] < dmin ? L.red[6] : dmin;
                dmin = L.red[7] < dmin ? L.red[7] : dmin;
                dmin = WR_UNIFORM(dmin);
                const int top = ak < mhi[0] ? ak : mhi[0];
                const int bottom = ak > mlo[0] ? ak : mlo[0];
                uint32_t enc = 0xffffffffu;
#pragma unroll
                for (int c = 0; c < NCW; c++) {
                    const bool keep = inr[c] && (dist[c] - dmin <= 50);
                    const uint32_t l16 = (keep && kc[c] < top) ? (uint32_t)jc[c] : 0xffffu;
                    const uint32_t h16 = (keep && kc[c] > bottom) ? (uint32_t)(W - 1 - jc[c]) : 0xffffu;
                    enc = wr_pk_min_u16(enc, l16 | (h16 << 16));
                }
                {
                    const uint32_t wred = WR_WAVE_PKMIN_U16(enc);
                    if (lane == 0) L.red[8 + wave] = (int32_t)wred;
                }
                WR_BARRIER(); // B2
                uint32_t red = wr_pk_min_u16(wr_pk_min_u16((uint32_t)L.red[8], (uint32_t)L.red[9]), wr_pk_min_u16((uint32_t)L.red[10], (uint32_t)L.red[11]));
                red = (uint32_t)WR_UNIFORM((int)red);
                int nlo = mlo[0], nhi = mhi[0];
                if (mlo[0] < top) nlo = (red & 0xffffu) != 0xffffu ? alo + (int)(red & 0xffffu) : top;
                if (mhi[0] > bottom) nhi = (red >> 16) != 0xffffu ? alo + (W - 1 - (int)(red >> 16)) : bottom;
                if (nlo != mlo[0] || nhi != mhi[0]) {
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
                    for (int c = 0; c < NCW; c++) {
                        const int slot = tid + T * c, k = kc[c];
                        if (inr[c] && (uint32_t)(k - nlo) > nmsp) rM[ms * W + slot] = WR_NULL_OFF;
                        if ((uint32_t)(k - oil) <= oisp && (uint32_t)(k - ilo[0]) > nisp) rI[is * W + slot] = WR_NULL_OFF;
                        if ((uint32_t)(k - odl) <= odsp && (uint32_t)(k - dlo[0]) > ndsp) rD[is * W + slot] = WR_NULL_OFF;
                    }
                    mlo[0] = nlo;
                    mhi[0] = nhi;
                }
            }
        }
        if (done) break;
        s += 2;
        if (s >= p.max_score) {
            status = 1;
            break;
        }
#pragma unroll
        for (int a = 4; a > 0; a--) {
            mlo[a] = mlo[a - 1];
            mhi[a] = mhi[a - 1];
        }
        ilo[1] = ilo[0];
        ihi[1] = ihi[0];
        dlo[1] = dlo[0];
        dhi[1] = dhi[0];
        ms = ms == 4 ? 0 : ms + 1;
        is ^= 1;
        // sources: M[s-4] (mismatch), M[s-8] (gap open), I[s-2] / D[s-2] (gap extension)
        int lo = mlo[2] < mlo[4] - 1 ? mlo[2] : mlo[4] - 1, hi = mhi[2] > mhi[4] + 1 ? mhi[2] : mhi[4] + 1;
        {
            const int l2 = ilo[1] + 1 < dlo[1] - 1 ? ilo[1] + 1 : dlo[1] - 1, h2 = ihi[1] + 1 > dhi[1] - 1 ? ihi[1] + 1 : dhi[1] - 1;
            lo = l2 < lo ? l2 : lo;
            hi = h2 > hi ? h2 : hi;
        }
        if (lo > hi) { // no source wavefront (all four empty)
            mlo[0] = ilo[0] = dlo[0] = E_LO;
            mhi[0] = ihi[0] = dhi[0] = E_HI;
#pragma unroll
            for (int c = 0; c < NCW; c++) {
                rM[ms * W + tid + T * c] = WR_NULL_OFF;
                rI[is * W + tid + T * c] = WR_NULL_OFF;
                rD[is * W + tid + T * c] = WR_NULL_OFF;
            }
            alo = 0;
            if (tid == 0) { // an empty row: same offset as the next one
                p.hdr2[s] = 0;
                p.hdr2[s + 1] = used;
                p.hdr2[s + 3] = used;
            }
            continue;
        }
        const int wd = hi - lo + 1;
        if (wd > W - 2) {
            status = 3;
            wide_at = wd;
            break;
        }
        if ((int64_t)used + wd > (int64_t)p.arena_cap) {
            status = 1;
            break;
        }
        const int32_t rowb = used;
        used += wd;
        alo = lo;
        if (tid == 0) { // entry s/2 = {lo, row offset}; the offset of entry s/2+1 closes the row
            p.hdr2[s] = lo;
            p.hdr2[s + 1] = rowb;
            p.hdr2[s + 3] = used;
        }
        const int r4 = ms >= 2 ? ms - 2 : ms + 3, r8 = ms == 4 ? 0 : ms + 1, r2 = is ^ 1; // rows of s-4, s-8, s-2
        WR_BARRIER(); // B3: every cell that left a range holds NULL, every ring row of the earlier scores is complete
        const uint32_t cmr = chunk_mask(lo, hi);
        int kk[NCW];
        int32_t vins[NCW], vdel[NCW], vmx[NCW];
        uint32_t em = 0xffffffffu, ei = 0xffffffffu, ed = 0xffffffffu; // (first, W-1-last) cell inside the DP matrix
#pragma unroll
        for (int c = 0; c < NCW; c++) {
            const int slot = tid + T * c;
            const int j = (slot - koff - lo) & (W - 1);
            const int k = lo + j;
            kk[c] = k;
            vins[c] = vdel[c] = vmx[c] = WR_NULL_OFF;
            if (((cmr >> c) & 1u) == 0) continue;
            const bool in = k <= hi;
            const int sm1 = (slot + W - 1) & (W - 1), sp1 = (slot + 1) & (W - 1);
            int32_t a = rM[r8 * W + sm1], b = rI[r2 * W + sm1];
            const bool iext = b >= a; // equal offsets: extension
            const int32_t ins = (iext ? b : a) + 1;
            a = rM[r8 * W + sp1];
            b = rD[r2 * W + sp1];
            const bool dext = b >= a;
            const int32_t del = dext ? b : a;
            const int32_t mis = rM[r4 * W + slot] + 1;
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
        {
            const uint32_t wm = WR_WAVE_PKMIN_U16(em), wi = WR_WAVE_PKMIN_U16(ei), wdl = WR_WAVE_PKMIN_U16(ed);
            if (lane == 0) {
                L.red[12 + wave] = (int32_t)wm;
                L.red[16 + wave] = (int32_t)wi;
                L.red[20 + wave] = (int32_t)wdl;
            }
        }
        WR_BARRIER(); // B4: the partial ranges; every thread has also read the old rows it needs
        {
            uint32_t rm = wr_pk_min_u16(wr_pk_min_u16((uint32_t)L.red[12], (uint32_t)L.red[13]), wr_pk_min_u16((uint32_t)L.red[14], (uint32_t)L.red[15]));
            uint32_t ri = wr_pk_min_u16(wr_pk_min_u16((uint32_t)L.red[16], (uint32_t)L.red[17]), wr_pk_min_u16((uint32_t)L.red[18], (uint32_t)L.red[19]));
            uint32_t rd = wr_pk_min_u16(wr_pk_min_u16((uint32_t)L.red[20], (uint32_t)L.red[21]), wr_pk_min_u16((uint32_t)L.red[22], (uint32_t)L.red[23]));
            rm = (uint32_t)WR_UNIFORM((int)rm);
            ri = (uint32_t)WR_UNIFORM((int)ri);
            rd = (uint32_t)WR_UNIFORM((int)rd);
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
        for (int c = 0; c < NCW; c++) { // (k - E_LO) as unsigned is above every span, also above the span of an empty range
            const int slot = tid + T * c, k = kk[c];
            rM[ms * W + slot] = (uint32_t)(k - mlo[0]) <= spm ? vmx[c] : WR_NULL_OFF;
            rI[is * W + slot] = (uint32_t)(k - ilo[0]) <= spi ? vins[c] : WR_NULL_OFF;
            rD[is * W + slot] = (uint32_t)(k - dlo[0]) <= spd ? vdel[c] : WR_NULL_OFF;
        }
    }
    if (WIN) { // a byte that is not A/C/G/T was packed on the way: the byte-comparing kernel takes the problem
        const bool wbad = WR_BALLOT(bad) != 0ull;
        WR_BARRIER();
        if (lane == 0) L.red[wave] = wbad ? 1 : 0;
        WR_BARRIER();
        if (status == 0 && (L.red[0] | L.red[1] | L.red[2] | L.red[3]) != 0) status = 3;
        status = WR_UNIFORM(status);
    }
    res->status = status;
    res->score = status == 0 ? s : (status == 3 ? wide_at : 0);
    res->used = used;
}
