// wfa_mw2_fwd.h - the forward pass of wfa_lean2_fwd.h for a WORKGROUP of four wavefronts per alignment
// (256 threads x NCW cells = 512 / 1024 diagonals): the restructuring of round 4's workgroup kernel (k_wfa_mw) that k_wfa_lean got.
// Equal to the oracle on the host SIMT emulator (tests/test_wfa_mw2_emulated_cpu.py) and on the GPU (tests/test_gpu_wfa_mw.py).
//
// k_wfa_mw is latency-bound: a handful of 20-50-kb alignments per round, each a chain of score steps, every round of the C3
// pipeline waits for them.  Its step has FOUR workgroup barriers (extension results; kept range of the cut-off; NULL-backs and
// old rows; trimmed ranges), five packed DPP reductions per wavefront, and every thread combines the four wavefronts'
// partial results from 4 LDS words per quantity.  Here:
//  * ring without wrap (frame recentred by the workgroup), lane order = diagonal order inside every group of 64 slots, so
//    a wavefront's first / last valid cells are ballots + scalar bit scans (no DPP reductions but the cut-off's minimum);
//  * the extension fused behind the recurrence (registers; one store of M, the cut-off applied);
//  * THREE barriers per score with the cut-off, two without: A - the wavefronts' partial ranges / end flag / minimum distance
//    (one 32-word strip of LDS: thirty-two lanes read one word each, a quad reduction, eight v_readlane), B - the kept range,
//    C - the rows of the score;
//  * rare events (end, limits, empty row, frame, scratch) in an outer loop.
// Same recurrence, tie rules, trimming, cut-off, backtrace bytes and header as k_wfa_lean / k_wfa_mw.  One source for the
// device and for the host emulator: every wave-level operation in wave-uniform, every barrier in workgroup-uniform control flow.
#pragma once
#include "lm_wfa_lean2_fwd.h"

#define MW2_THREADS 256
#define MW2_RED_WORDS 64 /* LDS words of reduction scratch */

// cells of the ring: nine rows of 256 * NCW cells + a pad cell on either side
template <int NCW> constexpr int mw2_ring_cells() { return 9 * (MW2_THREADS * NCW + 2); }

// the workgroup makes words [qw0, qw0 + WINW) of Q and [tw0, tw0 + WINW) of T resident (l2_win_move2 with barriers)
WR_DEV void mw2_win_move2(uint32_t *qbuf, const uint8_t *q, int plen, int *qw0_cur, int qw0, uint32_t *tbuf, const uint8_t *t, int tlen, int *tw0_cur,
                          int tw0, int tid, bool *bad, bool fresh) {
    WR_BARRIER(); // every thread is done reading the slots that are about to change
    const bool qkeep = !fresh && qw0 >= *qw0_cur && qw0 < *qw0_cur + L2_WINW, tkeep = !fresh && tw0 >= *tw0_cur && tw0 < *tw0_cur + L2_WINW;
    const int qfrom = qkeep ? *qw0_cur + L2_WINW : qw0, tfrom = tkeep ? *tw0_cur + L2_WINW : tw0;
    const int nq = qw0 + L2_WINW - qfrom, nt = tw0 + L2_WINW - tfrom;
    for (int i = tid; i < nq + nt; i += MW2_THREADS) {
        const bool isq = i < nq;
        const int w = isq ? qfrom + i : tfrom + (i - nq);
        const uint8_t *src = isq ? q : t;
        uint32_t *buf = isq ? qbuf : tbuf;
        const int nb = (isq ? plen : tlen) - 16 * w;
        const uint32_t word = nb > 0 ? l2_pack16(src + 16 * (int64_t)w, nb, bad) : 0u;
        const int slot = w & (L2_WINW - 1);
        buf[slot] = word;
        if (slot < 2) buf[L2_WINW + slot] = word;
    }
    *qw0_cur = WR_UNIFORM(qw0);
    *tw0_cur = WR_UNIFORM(tw0);
    WR_BARRIER();
}

// qb / tb: as in wfa_lean2_forward (whole packed sequences with a readable word in front, or - WIN - the two windows, filled
// here).  red: MW2_RED_WORDS words of LDS.  Results identical in every thread.
template <int NCW, bool WIN> WR_DEV void wfa_mw2_forward(const L2Prob &p, int32_t *ring, uint32_t *qb, uint32_t *tb, int32_t *red, L2Res *res) {
    static_assert(NCW == 1 || NCW == 2 || NCW == 4, "1, 2 or 4 cells per thread");
    constexpr int T = MW2_THREADS, W = T * NCW, RS = W + 2;
    constexpr int RNULL = WR_NULL_OFF;
    constexpr int E_LO = 1 << 28, E_HI = -(1 << 28), BIG = 1 << 29;
    const int tid = WR_TID, lane = tid & 63, wave = tid >> 6;
    const int plen = p.plen, tlen = p.tlen, ak = tlen - plen;
    int status = 0, wide_at = 0, nrec = 0;
    for (int i = tid; i < 9 * RS; i += T) ring[i] = RNULL;
    if (tid < MW2_RED_WORDS) red[tid] = BIG;
    // cell (row r, slot i) = ring[r * RS + 1 + i]; this thread's cell of chunk c is slot tid + 256 c = 64 G + lane, G = wave + 4 c
    int kbase = WR_UNIFORM(-(W / 2 - (ak >= -(W / 2 - 8) && ak <= W / 2 - 8 ? ak / 2 : 0))); // the band 0 .. ak centred on the ring
    int kcol[NCW];
#pragma unroll
    for (int c = 0; c < NCW; c++) kcol[c] = kbase + tid + T * c;
    int mlo[5], mhi[5], ilo[2], ihi[2], dlo[2], dhi[2];
#pragma unroll
    for (int a = 0; a < 5; a++) {
        mlo[a] = E_LO;
        mhi[a] = E_HI;
    }
#pragma unroll
    for (int a = 0; a < 2; a++) {
        ilo[a] = dlo[a] = E_LO;
        ihi[a] = dhi[a] = E_HI;
    }
    if (p.max_score < 1 || p.arena_cap < 1) status = 1;
    int s = 0;
    typedef WR_LDS int32_t *LP;
    LP pM[5], pI[2], pD[2]; // this thread's cell - 1 (chunk 0) in the ring rows by age (see wfa_lean2_forward)
#pragma unroll
    for (int a = 0; a < 5; a++) pM[a] = (LP)ring + tid + ((5 - a) % 5) * RS;
    pI[0] = (LP)ring + tid + 5 * RS;
    pI[1] = (LP)ring + tid + 6 * RS;
    pD[0] = (LP)ring + tid + 7 * RS;
    pD[1] = (LP)ring + tid + 8 * RS;
    int32_t used = 1;
    if (tid == 0) {
        p.hdr2[0] = 0;
        p.hdr2[1] = 0;
        p.hdr2[2] = 0;
        p.hdr2[3] = 1;
    }
    bool bad = false;
    int qw0 = 0, tw0 = 0;
    if (WIN)
        mw2_win_move2(qb, p.q, plen, &qw0, 0, tb, p.t, tlen, &tw0, 0, tid, &bad, true);
    else
        WR_BARRIER(); // the ring and the reduction strips are initialised
    // the minimum of one word per (quantity, wavefront) over the four wavefronts: strip[4 q + wave], q < 8 -> lanes 4 q .. 4 q + 3
    // read a word each, a quad reduction leaves the minimum of quantity q in those lanes
    auto strip_min = [&](const int32_t *strip) {
        const int32_t v = lane < 32 ? strip[lane] : BIG;
        return (int32_t)WR_QUAD_MIN_I32(v);
    };
    // greedy extension of whole-sequence cells (per wavefront; see wfa_lean2_forward)
    // INTERIOR / EDGE as in wfa_lean2_forward; a wavefront notes its own touches (touch_w), strip A carries them to all four
    int edge_m = 0, touch_w = 0;
    auto extend = [&](auto in_edge, bool valid, int h, int k) {
        const int hmax = tlen < plen + k ? tlen : plen + k;
        int lim = valid ? hmax : 0;
        while (true) {
            const bool ext = h < lim;
            if (WR_BALLOT(ext) == 0ull) break;
            const uint32_t d = l2_get16(qb, h - k) ^ l2_get16(tb, h);
            const int nm = WR_CLZ(d) >> 1;
            h += ext ? nm : 0;
            lim = nm == 16 ? lim : h;
        }
        if (!decltype(in_edge)::value) touch_w = WR_UNIFORM(touch_w | (WR_BALLOT(valid && h >= hmax) != 0ull ? -1 : 0));
        return h < hmax ? h : hmax;
    };
    // WIN: all cells of the workgroup through the windows; cells outside a window wait, the workgroup moves both windows to
    // the smallest waiting positions (strip red[48 .. 59]: waiting flag, smallest v, smallest h per wavefront)
    auto extend_win = [&](auto in_edge, int *h, const int *k, const bool *valid, const bool *on) {
        int lim[NCW], hmax[NCW];
#pragma unroll
        for (int c = 0; c < NCW; c++) {
            hmax[c] = tlen < plen + k[c] ? tlen : plen + k[c];
            lim[c] = valid[c] ? hmax[c] : 0;
        }
        while (true) {
            uint64_t pend = 0;
#pragma unroll
            for (int c = 0; c < NCW; c++) {
                if (!on[c]) continue; // (wave-uniform)
                while (true) {
                    const uint64_t go_m = WR_BALLOT(h[c] < lim[c]) & WR_BALLOT(l2_win_has(qw0, h[c] - k[c])) & WR_BALLOT(l2_win_has(tw0, h[c]));
                    if (go_m == 0ull) break;
                    uint32_t qh, ql, th, tl;
                    l2_win_get32(qb, h[c] - k[c], &qh, &ql);
                    l2_win_get32(tb, h[c], &th, &tl);
                    const uint32_t dh = qh ^ th, dl = ql ^ tl;
                    const int nm = dh ? WR_CLZ(dh) >> 1 : 16 + (WR_CLZ(dl) >> 1);
                    const bool go = h[c] < lim[c] && l2_win_has(qw0, h[c] - k[c]) && l2_win_has(tw0, h[c]);
                    h[c] += go ? nm : 0;
                    lim[c] = (!go || nm == 32) ? lim[c] : h[c];
                }
                pend |= WR_BALLOT(h[c] < lim[c]);
            }
            int mv = BIG, mh = BIG;
#pragma unroll
            for (int c = 0; c < NCW; c++) {
                const bool wt = on[c] && h[c] < lim[c];
                mh = wt && h[c] < mh ? h[c] : mh;
                mv = wt && h[c] - k[c] < mv ? h[c] - k[c] : mv;
            }
            mv = WR_WAVE_MIN_I32(mv);
            mh = WR_WAVE_MIN_I32(mh);
            if (lane == 0) {
                red[48 + wave] = pend ? -1 : 0;
                red[52 + wave] = mv;
                red[56 + wave] = mh;
            }
            WR_BARRIER();
            const int32_t r = strip_min(red + 48); // lanes 0-3: anybody waiting, 4-7: smallest v, 8-11: smallest h
            const int any = WR_READLANE(r, 0), gv = WR_READLANE(r, 4), gh = WR_READLANE(r, 8);
            if (any == 0) {
                WR_BARRIER(); // (the strip may be written again)
                break;
            }
            mw2_win_move2(qb, p.q, plen, &qw0, gv >> 4, tb, p.t, tlen, &tw0, gh >> 4, tid, &bad, false);
        }
#pragma unroll
        for (int c = 0; c < NCW; c++) {
            if (!decltype(in_edge)::value && on[c]) touch_w = WR_UNIFORM(touch_w | (WR_BALLOT(valid[c] && h[c] >= hmax[c]) != 0ull ? -1 : 0));
            h[c] = h[c] < hmax[c] ? h[c] : hmax[c];
        }
    };
    bool done = false;
    if (status == 0) { // score 0: the cell of diagonal 0 (its slot is W / 2 - ak / 2)
        const bool mine = kcol[0] == 0;   // (chunk 0 holds slots 0 .. 255; W / 2 is in chunk NCW / 2: test every chunk)
        int h0[NCW], k0[NCW];
        bool v0[NCW], on0[NCW];
        bool any_mine = false;
#pragma unroll
        for (int c = 0; c < NCW; c++) {
            v0[c] = kcol[c] == 0;
            on0[c] = WR_BALLOT(v0[c]) != 0ull;
            h0[c] = k0[c] = 0;
            any_mine = any_mine || v0[c];
        }
        (void)mine;
        if (WIN) {
            extend_win(std::false_type{}, h0, k0, v0, on0);
        } else {
#pragma unroll
            for (int c = 0; c < NCW; c++)
                if (on0[c]) h0[c] = extend(std::false_type{}, v0[c], 0, 0);
        }
        int hh = 0;
#pragma unroll
        for (int c = 0; c < NCW; c++) {
            if (v0[c]) pM[0][T * c + 1] = h0[c];
            hh = v0[c] ? h0[c] : hh;
        }
        mlo[0] = mhi[0] = 0;
        // the end already ?  (the thread of diagonal 0 says so through the strip)
        if (any_mine) {
            red[0] = (ak == 0 && hh >= tlen) ? -1 : 0;
            red[1] = hh >= (tlen < plen ? tlen : plen) ? -1 : 0; // the first touch already ?
        }
        WR_BARRIER();
        done = WR_UNIFORM(red[0]) != 0;
        edge_m = WR_UNIFORM(red[1]);
        touch_w = 0;
        WR_BARRIER();
        if (tid == 0) red[0] = red[1] = BIG;
        WR_BARRIER();
    }
    const int s_limit = p.max_score;
    int s_lim = s_limit;
    int shrink_from = 0;
    int lo = 0, hi = 0; // the row of score s + 2 as the hot loop saw it when it left
    auto hot = [&](auto in_edge) {
        constexpr bool EDGE = decltype(in_edge)::value;
        while (true) {
            lo = WR_UNIFORM(mlo[1] < mlo[3] - 1 ? mlo[1] : mlo[3] - 1);
            hi = WR_UNIFORM(mhi[1] > mhi[3] + 1 ? mhi[1] : mhi[3] + 1);
            {
                const int l2 = WR_UNIFORM(ilo[0] + 1 < dlo[0] - 1 ? ilo[0] + 1 : dlo[0] - 1), h2 = WR_UNIFORM(ihi[0] + 1 > dhi[0] - 1 ? ihi[0] + 1 : dhi[0] - 1);
                lo = l2 < lo ? l2 : lo;
                hi = h2 > hi ? h2 : hi;
            }
            const uint32_t span = (uint32_t)(hi - lo);
            const int gf = (lo - kbase) >> 6, gl = (hi - kbase) >> 6; // groups of 64 slots holding cells of [lo, hi]
            uint32_t rare = (EDGE ? 0u : (uint32_t)edge_m) | (uint32_t)(s_lim - 3 - s) | span | (uint32_t)(lo - kbase) | (uint32_t)(kbase + W - 1 - hi) |
                            ((uint32_t)p.arena_cap - (uint32_t)used - span - 1u);
            // (fewer chunks: a chunk here is 256 slots = one cell of every thread)
            if (NCW > 1)
                rare |= (uint32_t)((int)((span + 1 + 2 * L2_SHRINK_MARGIN + T - 1) / T) - ((int)((uint32_t)(hi - kbase) / T) - (int)((uint32_t)(lo - kbase) / T) + 1)) &
                        ~(uint32_t)(s + 2 - WR_UNIFORM(shrink_from));
            if ((int32_t)rare < 0) break;
            // ---- a plain step ----
            s += 2;
#pragma unroll
            for (int a = 4; a > 0; a--) {
                mlo[a] = mlo[a - 1];
                mhi[a] = mhi[a - 1];
            }
            ilo[1] = ilo[0];
            ihi[1] = ihi[0];
            dlo[1] = dlo[0];
            dhi[1] = dhi[0];
            {
                const LP t = pM[4];
#pragma unroll
                for (int a = 4; a > 0; a--) pM[a] = pM[a - 1];
                pM[0] = t;
                const LP ti = pI[0], td = pD[0];
                pI[0] = pI[1];
                pI[1] = ti;
                pD[0] = pD[1];
                pD[1] = td;
            }
            const LP newM = pM[0], newI = pI[0], newD = pD[0];
            const int32_t rowb = used;
            used += (int32_t)span + 1;
            if (tid == 0) {
                p.hdr2[s] = lo;
                p.hdr2[s + 1] = rowb;
                p.hdr2[s + 3] = used;
            }
            const LP M8 = pM[4], M4 = pM[2], I2 = pI[1], D2 = pD[1];
            const int32_t rowk = rowb - lo;
            int32_t off[NCW], vins[NCW], vdel[NCW];
            // this wavefront's first / last cell inside the DP matrix per wavefront kind, as slots; "none" = BIG / -BIG
            int fm = BIG, fi = BIG, fd = BIG, lm = -BIG, li = -BIG, ld = -BIG;
            uint32_t cmv = 0;
#pragma unroll
            for (int c = 0; c < NCW; c++) {
                off[c] = RNULL;
                vins[c] = vdel[c] = RNULL;
                const int G = wave + 4 * c; // (wave-uniform)
                if (G < gf || G > gl) continue;
                const int k = kcol[c];
                int32_t a = M8[T * c], b = I2[T * c];
                const bool iext = b >= a;
                const int32_t ins = (iext ? b : a) + 1;
                a = M8[T * c + 2];
                b = D2[T * c + 2];
                const bool dext = b >= a;
                const int32_t del = dext ? b : a;
                const int32_t mis = M4[T * c + 1] + 1;
                int32_t mx = mis > ins ? mis : ins;
                if (del > mx) mx = del;
                const uint32_t mc = (mis >= del && mis >= ins) ? 0u : (del >= ins ? 2u : 1u);
                if ((uint32_t)(k - lo) <= span) p.bt[(uint32_t)(rowk + k)] = (uint8_t)(mc | (iext ? 4u : 0u) | (dext ? 8u : 0u));
                vins[c] = ins;
                vdel[c] = del;
                if (!EDGE) { // interior: a cell is valid when a source is; the row is [lo, hi] and every active group holds cells of it
                    off[c] = mx < 0 ? RNULL : mx;
                    cmv |= 1u << c;
                    continue;
                }
                if ((uint32_t)mx > (uint32_t)tlen) mx = RNULL;
                if ((uint32_t)(mx - k) > (uint32_t)plen) mx = RNULL;
                off[c] = mx;
                const uint64_t bm = WR_BALLOT(mx >= 0);
                const uint64_t bi = WR_BALLOT((uint32_t)ins <= (uint32_t)tlen) & WR_BALLOT((uint32_t)(ins - k) <= (uint32_t)plen);
                const uint64_t bd = WR_BALLOT((uint32_t)del <= (uint32_t)tlen) & WR_BALLOT((uint32_t)(del - k) <= (uint32_t)plen);
                if (bm) {
                    const int f = 64 * G + WR_FF1(bm), l = 64 * G + 63 - WR_FLB(bm);
                    fm = f < fm ? f : fm;
                    lm = l > lm ? l : lm;
                    cmv |= 1u << c;
                }
                if (bi) {
                    const int f = 64 * G + WR_FF1(bi), l = 64 * G + 63 - WR_FLB(bi);
                    fi = f < fi ? f : fi;
                    li = l > li ? l : li;
                }
                if (bd) {
                    const int f = 64 * G + WR_FF1(bd), l = 64 * G + 63 - WR_FLB(bd);
                    fd = f < fd ? f : fd;
                    ld = l > ld ? l : ld;
                }
            }
            // ---- the new M cells, still in registers: greedy extension (per wavefront; WIN: the workgroup moves the windows) ----
            if (WIN) {
                int h_[NCW], k_[NCW];
                bool v_[NCW], on_[NCW];
#pragma unroll
                for (int c = 0; c < NCW; c++) {
                    v_[c] = off[c] >= 0;
                    on_[c] = ((cmv >> c) & 1u) != 0;
                    h_[c] = v_[c] ? off[c] : 0;
                    k_[c] = v_[c] ? kcol[c] : 0;
                }
                extend_win(in_edge, h_, k_, v_, on_);
#pragma unroll
                for (int c = 0; c < NCW; c++) off[c] = v_[c] ? h_[c] : RNULL;
            } else {
#pragma unroll
                for (int c = 0; c < NCW; c++) {
                    if (!((cmv >> c) & 1u)) continue;
                    const bool valid = off[c] >= 0;
                    const int h = extend(in_edge, valid, valid ? off[c] : 0, valid ? kcol[c] : 0);
                    off[c] = valid ? h : RNULL;
                }
            }
            // this wavefront's share of: the end (the cell of the final diagonal at the end of the target), the smallest distance
            int32_t dist[NCW];
            int32_t dm = BIG;
            bool fin = false;
#pragma unroll
            for (int c = 0; c < NCW; c++) {
                dist[c] = BIG;
                if (!((cmv >> c) & 1u)) continue;
                const int32_t lv = plen - off[c] + kcol[c], lh = tlen - off[c];
                dist[c] = lv > lh ? lv : lh;
                dist[c] = off[c] >= 0 ? dist[c] : BIG;
                dm = dist[c] < dm ? dist[c] : dm;
                fin = fin || (kcol[c] == ak && off[c] >= tlen);
            }
            {
                const int32_t wdm = cmv ? (int32_t)WR_WAVE_MIN_I32(dm) : BIG;
                const bool wfin = WR_BALLOT(fin) != 0ull;
                if (lane == 0) { // strip A: eight quantities x four wavefronts, all combined by a minimum
                    if (EDGE) {
                        red[0 + wave] = fm;
                        red[4 + wave] = -lm;
                        red[8 + wave] = fi;
                        red[12 + wave] = -li;
                        red[16 + wave] = fd;
                        red[20 + wave] = -ld;
                        red[24 + wave] = wfin ? -1 : 0;
                    } else {
                        red[20 + wave] = touch_w; // (interior: the first touch of an end, in a slot the ranges do not need)
                        red[24 + wave] = wfin ? -1 : 0; // (the first touch may be the end itself)
                    }
                    red[28 + wave] = wdm;
                }
            }
            WR_BARRIER(); // A: the partial results of the four wavefronts
            bool cut = false;
            {
                const int32_t r = strip_min(red);
                const int gfin = WR_READLANE(r, 24), dmin = WR_READLANE(r, 28);
                if (EDGE) {
                    const int gfm = WR_READLANE(r, 0), glm = -WR_READLANE(r, 4), gfi = WR_READLANE(r, 8), gli = -WR_READLANE(r, 12);
                    const int gfd = WR_READLANE(r, 16), gld = -WR_READLANE(r, 20);
                    mlo[0] = WR_UNIFORM(glm >= 0 ? kbase + gfm : E_LO);
                    mhi[0] = WR_UNIFORM(glm >= 0 ? kbase + glm : E_HI);
                    ilo[0] = WR_UNIFORM(gli >= 0 ? kbase + gfi : E_LO);
                    ihi[0] = WR_UNIFORM(gli >= 0 ? kbase + gli : E_HI);
                    dlo[0] = WR_UNIFORM(gld >= 0 ? kbase + gfd : E_LO);
                    dhi[0] = WR_UNIFORM(gld >= 0 ? kbase + gld : E_HI);
                    // the end test of k_wfa_lean: the cell of diagonal ak inside the M range
                    done = gfin != 0 && ak >= mlo[0] && ak <= mhi[0];
                    s_lim = WR_UNIFORM(done ? -(1 << 30) : s_lim);
                } else {
                    // interior (see wfa_lean2_forward): the ranges of the new wavefronts from those of their sources
                    const int i_lo = WR_UNIFORM((mlo[4] < ilo[1] ? mlo[4] : ilo[1]) + 1), i_hi = WR_UNIFORM((mhi[4] > ihi[1] ? mhi[4] : ihi[1]) + 1);
                    const int d_lo = WR_UNIFORM((mlo[4] < dlo[1] ? mlo[4] : dlo[1]) - 1), d_hi = WR_UNIFORM((mhi[4] > dhi[1] ? mhi[4] : dhi[1]) - 1);
                    ilo[0] = i_lo <= i_hi ? i_lo : E_LO;
                    ihi[0] = i_lo <= i_hi ? i_hi : E_HI;
                    dlo[0] = d_lo <= d_hi ? d_lo : E_LO;
                    dhi[0] = d_lo <= d_hi ? d_hi : E_HI;
                    mlo[0] = lo; // (= min / max over M[s-4], I[s], D[s]: the row itself)
                    mhi[0] = hi;
                    edge_m = WR_UNIFORM(WR_READLANE(r, 20)); // -1 once any wavefront's cell has touched an end
                    done = gfin != 0 && ak >= lo && ak <= hi;
                    s_lim = WR_UNIFORM(done ? -(1 << 30) : s_lim);
                }
                if (mhi[0] - mlo[0] + 1 >= 10) { // wf-adaptive(10, 50)  (workgroup-uniform)
                    const int top = ak < mhi[0] ? ak : mhi[0];
                    const int bottom = ak > mlo[0] ? ak : mlo[0];
                    int fl = BIG, lh_ = -BIG;
#pragma unroll
                    for (int c = 0; c < NCW; c++) {
                        if (!((cmv >> c) & 1u)) continue;
                        const int G = wave + 4 * c;
                        const uint64_t keep = WR_BALLOT(dist[c] - dmin <= 50);
                        const uint64_t kl = keep & WR_BALLOT(kcol[c] < top), kh = keep & WR_BALLOT(kcol[c] > bottom);
                        if (kl) {
                            const int f = 64 * G + WR_FF1(kl);
                            fl = f < fl ? f : fl;
                        }
                        if (kh) {
                            const int l = 64 * G + 63 - WR_FLB(kh);
                            lh_ = l > lh_ ? l : lh_;
                        }
                    }
                    if (lane == 0) { // strip B
                        red[32 + wave] = fl;
                        red[36 + wave] = -lh_;
                    }
                    WR_BARRIER(); // B: the kept range
                    const int32_t r2 = strip_min(red + 32);
                    const int gfl = WR_READLANE(r2, 0), glh = -WR_READLANE(r2, 4);
                    int nlo = mlo[0], nhi = mhi[0];
                    if (mlo[0] < top) nlo = gfl < BIG ? kbase + gfl : top;
                    if (mhi[0] > bottom) nhi = glh > -BIG ? kbase + glh : bottom;
                    if (nlo != mlo[0] || nhi != mhi[0]) {
                        cut = true;
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
                        mlo[0] = WR_UNIFORM(nlo);
                        mhi[0] = WR_UNIFORM(nhi);
                        ilo[0] = WR_UNIFORM(ilo[0]);
                        ihi[0] = WR_UNIFORM(ihi[0]);
                        dlo[0] = WR_UNIFORM(dlo[0]);
                        dhi[0] = WR_UNIFORM(dhi[0]);
                    }
                }
            }
            // ---- the three rows of score s: cells outside a range are NULL ----
            {
                const uint32_t spm = (uint32_t)(mhi[0] - mlo[0]), spi = (uint32_t)(ihi[0] - ilo[0]), spd = (uint32_t)(dhi[0] - dlo[0]);
#pragma unroll
                for (int c = 0; c < NCW; c++) {
                    const int k = kcol[c];
                    int32_t m = off[c];
                    if (cut) m = (uint32_t)(k - mlo[0]) <= spm ? m : RNULL;
                    newM[T * c + 1] = m;
                    newI[T * c + 1] = (uint32_t)(k - ilo[0]) <= spi ? vins[c] : RNULL;
                    newD[T * c + 1] = (uint32_t)(k - dlo[0]) <= spd ? vdel[c] : RNULL;
                }
            }
            WR_BARRIER(); // C: the rows of score s are in the ring (and both strips may be written again)
        }
    };
    while (status == 0 && !done) {
        if (edge_m)
            hot(std::true_type{});
        else
            hot(std::false_type{});
        // ---- what is due instead of a plain step (lo, hi: the row of score s + 2; possibly nothing but the first touch) ----
        if (done) break;
        if (s + 2 >= s_limit) {
            status = 1;
            break;
        }
        if (lo > hi) { // an empty row
            s += 2;
#pragma unroll
            for (int a = 4; a > 0; a--) {
                mlo[a] = mlo[a - 1];
                mhi[a] = mhi[a - 1];
            }
            ilo[1] = ilo[0];
            ihi[1] = ihi[0];
            dlo[1] = dlo[0];
            dhi[1] = dhi[0];
            {
                const LP t = pM[4];
#pragma unroll
                for (int a = 4; a > 0; a--) pM[a] = pM[a - 1];
                pM[0] = t;
                const LP ti = pI[0], td = pD[0];
                pI[0] = pI[1];
                pI[1] = ti;
                pD[0] = pD[1];
                pD[1] = td;
            }
            mlo[0] = ilo[0] = dlo[0] = E_LO;
            mhi[0] = ihi[0] = dhi[0] = E_HI;
#pragma unroll
            for (int c = 0; c < NCW; c++) pM[0][T * c + 1] = pI[0][T * c + 1] = pD[0][T * c + 1] = RNULL;
            if (tid == 0) {
                p.hdr2[s] = 0;
                p.hdr2[s + 1] = used;
                p.hdr2[s + 3] = used;
            }
            WR_BARRIER();
            continue;
        }
        if ((int64_t)used + (hi - lo + 1) > (int64_t)p.arena_cap) {
            status = 1;
            break;
        }
        {
            int ulo = lo < mlo[0] ? lo : mlo[0], uhi = hi > mhi[0] ? hi : mhi[0];
            ulo = mlo[2] < ulo ? mlo[2] : ulo;
            uhi = mhi[2] > uhi ? mhi[2] : uhi;
            ulo = ilo[0] < ulo ? ilo[0] : ulo;
            uhi = dhi[0] > uhi ? dhi[0] : uhi;
            const int uw = uhi - ulo + 1;
            if (uw > W) {
                status = 3;
                wide_at = uw;
                break;
            }
            int nch = (uw + 2 * L2_SHRINK_MARGIN + T - 1) / T; // chunks (of 256 slots) the live rows get
            nch = nch < NCW ? nch : NCW;
            const bool out = lo < kbase || hi > kbase + W - 1;
            const int touched = (int)((uint32_t)(uhi - kbase) / T) - (int)((uint32_t)(ulo - kbase) / T) + 1;
            if (!out && nch >= touched) {
                shrink_from = WR_UNIFORM(s + 2 + 8);
                continue;
            }
            const int nk = WR_UNIFORM(ulo - (T * nch - uw) / 2);
            const int delta = nk - kbase;
#pragma unroll 1
            for (int r = 0; r < 9; r++) {
                int32_t v[NCW];
#pragma unroll
                for (int c = 0; c < NCW; c++) {
                    const int src = tid + T * c + delta;
                    v[c] = (uint32_t)src < (uint32_t)W ? ring[r * RS + 1 + src] : RNULL;
                }
                WR_BARRIER(); // every thread has read the row
#pragma unroll
                for (int c = 0; c < NCW; c++) ring[r * RS + 1 + tid + T * c] = v[c];
            }
            WR_BARRIER();
            kbase = nk;
#pragma unroll
            for (int c = 0; c < NCW; c++) kcol[c] = nk + tid + T * c;
            nrec++;
        }
    }
    if (WIN) { // a byte that is not A / C / G / T was packed on the way
        const bool wbad = WR_BALLOT(bad) != 0ull;
        WR_BARRIER();
        if (lane == 0) red[48 + wave] = wbad ? -1 : 0;
        WR_BARRIER();
        if (status == 0 && WR_UNIFORM(red[48] | red[49] | red[50] | red[51]) != 0) status = 3;
    }
    res->qw0 = qw0;
    res->tw0 = tw0;
    res->status = status;
    res->score = status == 0 ? s : wide_at;
    res->used = used;
    res->recentres = nrec;
}
