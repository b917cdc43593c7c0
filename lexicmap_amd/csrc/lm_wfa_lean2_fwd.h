// wfa_lean2_fwd.h - the forward pass of the gap-affine WFA (x=4, o=6, e=2, wf-adaptive(10,50)) by ONE
// wavefront, restructured for fewer instructions per score step than k_wfa_lean (lexicmap_amd/csrc/lm_kernels.hip).
// Equal to the oracle on the host SIMT emulator (tests/test_wfa_lean2_emulated_cpu.py) and on the GPU (tests/test_gpu_wfa_lean2.py);
// instruction counts of the score loop: DESIGN.md section 4.  (k_wfa_lean below = its predecessor, removed in round 5.)
//
// Why: k_wfa_lean is 58 % of the vector and 71 % of the scalar instructions of a C3 step (profiles/r04_c3_pmc_sq.json) and it
// runs at its instruction roofline (experiments/README.md, valu_rate; round 4): only fewer instructions per score step make it faster.
// Its ISA spends more than half of a step on bookkeeping: the slot -> diagonal mapping of a ring that wraps (k, j, in-range
// masks per chunk and phase, twice per score), three packed DPP reductions for the trimmed ranges and two for the cut-off
// (6 dependent DPP stages each), the extension as a second phase that reads M back from LDS and writes it again, separate
// conditional stores that put NULL back after the cut-off, and scalar registers spilled to vector lanes.
//
// What changes (same recurrence, tie rules, trimming, cut-off, backtrace bytes and header as k_wfa_lean: its bt_walk /
// bt_replay read the rows written here):
//  * NO WRAP: the cell of diagonal k lives at slot k - kbase of every ring row (kbase wave-uniform), a row has a NULL pad cell
//    on either side.  The diagonal of a lane's cell is a loop-invariant register, the neighbours k-1 / k+1 are the same
//    address +- one cell (an instruction offset), and lane order = diagonal order.  When the live rows drift out of the frame
//    (every few thousand scores: two indels of the same kind per hundred bases) or fit in fewer 64-slot chunks than they
//    touch, all nine rows are shifted in LDS (recentre).
//  * lane order = diagonal order, so the first / last valid cell of a wavefront is a BALLOT (the compare that decides
//    validity already is one) + s_ff1 / s_flbit on the scalar unit: no DPP reduction for M, I, D or for the kept range of
//    the cut-off (its minimum distance still is one).
//  * the extension of M[s] is FUSED behind its computation: the new offsets are extended in registers and stored once, with
//    the cut-off already applied - no read-back, no second store, no NULL-back stores, one LDS hand-off per score instead of
//    three.  (M cells outside the trimmed range are NULL by sanitisation, so "valid" is all the extension has to know.)
//  * extension step: v_alignbit_b32 on the two words that hold the last of the 32 bits, one running offset per cell, the
//    end-of-sequence clamp hoisted out of the loop (hmax per cell), the predicate recomputed from registers every pass.
//  * INTERIOR / EDGE: until some M cell has reached the end of a sequence every cell with a valid source is inside the DP
//    matrix and the trimmed ranges are scalar minima / maxima of the older ranges; two copies of the hot loop.
//  * the hot loop is the plain score step only (one sign test decides); rare events live in an outer loop; the lane's row
//    addresses rotate in vector registers instead of "row = f(s mod 5)" on the scalar unit.
// One source for the device (hipcc) and for the host emulator (tests/emu/simt_emu.h): every cross-lane operation sits in
// wave-uniform control flow.
#pragma once
#include <stdint.h>

#include <type_traits>

#ifndef WR_NULL_OFF
#define WR_NULL_OFF (-1073741824) /* = LM_NULL_OFF */
#endif
#ifndef L2_SHRINK_MARGIN
#define L2_SHRINK_MARGIN 8 /* a narrower flavour only when the live rows fit its chunks with this many free slots on either side */
#endif

struct L2Prob {        // one alignment (wave-uniform)
    const uint8_t *q, *t; // the ASCII sequences (read by the sliding-window form only)
    int32_t plen, tlen;
    int32_t *hdr2;     // {first diagonal, row offset} per even score (max_score + 4 entries)
    uint8_t *bt;       // backtrace bytes
    int32_t arena_cap; // bytes usable at bt
    int32_t max_score;
};
struct L2Res {
    int32_t status; // 0 aligned, 1 scratch / score overflow, 3 live rows wider than the ring
    int32_t score;  // final score (status 0) or the width that did not fit (status 3)
    int32_t used;   // backtrace bytes written
    int32_t recentres;
    int32_t qw0, tw0; // WIN: first resident word of either window when the pass ended (bt_replay continues from there)
};

// cells of the ring: nine rows (M 0-4, I 5-6, D 7-8) of 64 * NC cells + a pad cell on either side
template <int NC> constexpr int l2_ring_cells() { return 9 * (64 * NC + 2); }

// WR_NEG2(pos): a shift count equal to -2 pos modulo 32.  (Written in C - in whatever form - the compiler reduces it to
// v_mul_lo_u32 pos, 30: the funnel shift only looks at five bits, and 30 = -2 modulo 32.  That is a quarter-rate instruction,
// twice per extension pass and chunk; the device macro is one full-rate v_mul_u32_u24.)
#ifndef WR_NEG2
#define WR_NEG2(pos) (30u * (uint32_t)(pos))
#endif
// 16 packed bases from base `pos` (first base in the top bits of a word; seq[-1] must be readable): the two words that hold
// the LAST of the 32 bits, funnel-shifted - one v_alignbit_b32 whatever the position, no 64-bit shift, no half swaps
WR_DEV uint32_t l2_get16(const uint32_t *seq, int pos) {
    const uint32_t *w = seq + ((pos + 15) >> 4); // = (2 pos + 31) >> 5: the word of the last bit
    return WR_ALIGNBIT(w[-1], w[0], WR_NEG2(pos)); // ({w[-1], w[0]} >> (-2 pos & 31)) & 0xffffffff
}

// ---- sliding 2-bit windows (the WIN form): the layout of k_wfa_lean's WfaWin - each sequence a circular window of L2_WINW
// words of 16 bases, word w at slot w & (L2_WINW - 1), slots 0 and 1 mirrored behind the last one so that three consecutive
// words never wrap; bt_replay<true> continues on the same buffers.  buf[-1] must be readable.
#define L2_WINW 256
WR_DEV uint32_t l2_pack_base(uint32_t c, bool *bad) {
    const uint32_t code = (c >> 1) & 3u;
    *bad |= c != ((0x47544341u >> (code << 3)) & 0xffu); // 'A','C','T','G' by code
    return code;
}
WR_DEV uint32_t l2_pack16(const uint8_t *s, int nb, bool *bad) {
    uint32_t w = 0;
    if (nb >= 16) {
        uint32_t b[4];
        __builtin_memcpy(b, s, 16);
#pragma unroll
        for (int j = 0; j < 16; j++) w = (w << 2) | l2_pack_base((b[j >> 2] >> ((j & 3) << 3)) & 0xffu, bad);
    } else {
        for (int j = 0; j < nb; j++) w = (w << 2) | l2_pack_base(s[j], bad);
        w <<= 2 * (16 - nb);
    }
    return w;
}
// can 32 bases from `pos` be read from the window ? (words pos >> 4 .. +2)
WR_DEV bool l2_win_has(int w0, int pos) { return (uint32_t)((pos >> 4) - w0) < (uint32_t)(L2_WINW - 2); }
// 32 packed bases from base `pos`, as (first 16, next 16): three consecutive slots, two funnel shifts (see l2_get16; the slot
// in front of the window's first word is read and ignored when pos is a multiple of 16)
WR_DEV void l2_win_get32(const uint32_t *buf, int pos, uint32_t *hi, uint32_t *lo) {
    const uint32_t *w = buf + ((((pos + 15) >> 4) - 1) & (L2_WINW - 1));
    const uint32_t sh = WR_NEG2(pos);
    *hi = WR_ALIGNBIT(w[0], w[1], sh);
    *lo = WR_ALIGNBIT(w[1], w[2], sh);
}
// the wavefront makes words [qw0, qw0 + WINW) of Q and [tw0, tw0 + WINW) of T resident; words that stay are not reloaded
WR_DEV void l2_win_move2(uint32_t *qbuf, const uint8_t *q, int plen, int *qw0_cur, int qw0, uint32_t *tbuf, const uint8_t *t, int tlen, int *tw0_cur,
                         int tw0, int lane, bool *bad, bool fresh) {
    WR_WAVE_SYNC(); // every lane is done reading the slots that are about to change
    const bool qkeep = !fresh && qw0 >= *qw0_cur && qw0 < *qw0_cur + L2_WINW, tkeep = !fresh && tw0 >= *tw0_cur && tw0 < *tw0_cur + L2_WINW;
    const int qfrom = qkeep ? *qw0_cur + L2_WINW : qw0, tfrom = tkeep ? *tw0_cur + L2_WINW : tw0;
    const int nq = qw0 + L2_WINW - qfrom, nt = tw0 + L2_WINW - tfrom; // words to load (0 when a window does not move)
    for (int i = lane; i < nq + nt; i += 64) {
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
    WR_WAVE_SYNC();
}

#ifndef L2_FWD_ATTR
#define L2_FWD_ATTR WR_DEV
#endif
#ifndef L2_COUNT
#define L2_COUNT(what, n) /* the emulator harness counts score steps per flavour, extension passes and cut-offs */
#endif
// qb / tb: word 0 of the 2-bit packed sequences in LDS (one readable word in front, (len + 15) / 16 + 2 words, zero behind the
// last base); WIN: the two windows instead (L2_WINW + 2 words each, one readable word in front), filled here from p.q / p.t
template <int NC, typename RT, bool WIN = false, int MARGIN = L2_SHRINK_MARGIN>
L2_FWD_ATTR void wfa_lean2_forward(const L2Prob &p, RT *ring, uint32_t *qb, uint32_t *tb, L2Res *res) {
    static_assert(NC == 1 || NC == 2 || NC == 4 || NC == 8 || NC == 16, "1, 2, 4, 8 or 16 cells per lane");
    static_assert(!(WIN && sizeof(RT) == 2), "16-bit cells: whole sequences of at most 12 000 bases");
    constexpr int W = 64 * NC, RS = W + 2;
    constexpr bool R16 = sizeof(RT) == 2;
    constexpr int RNULL = R16 ? -16384 : WR_NULL_OFF; // (16-bit cells: k_wfa_lean's argument - sequences <= 12000, s < 24000)
    constexpr int E_LO = 1 << 28, E_HI = -(1 << 28);
    const int lane = WR_TID & 63;
    const int plen = p.plen, tlen = p.tlen, ak = tlen - plen;
    int status = 0, wide_at = 0, nrec = 0;
    for (int i = lane; i < 9 * RS; i += 64) ring[i] = (RT)RNULL;
    // cell (row r, slot i) = ring[r * RS + 1 + i]; this lane's cell of chunk c is slot lane + 64 c
    RT *const cell0 = ring + 1 + lane;
    // the band between diagonal 0 and the final diagonal starts centred on the first chunk
    int kbase = WR_UNIFORM(-(32 - (ak >= -40 && ak <= 40 ? ak / 2 : 0)));
    int kcol[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) kcol[c] = kbase + lane + 64 * c;
    // valid ranges by age in even scores: mlo[a]..mhi[a] is M[s-2a]; an empty range is (E_LO, E_HI)
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
    // FLAVOURS (round 6).  Every live row lies in chunks [0, na) of the frame - the ring rows are all NULL beyond, since the shift
    // that last centred them rewrote every cell - and the hot loop exists in one copy per flavour NA (up to three: NA_MIN, 2 NA_MIN,
    // 4 NA_MIN <= NC) that works on chunks 0 .. NA-1 UNCONDITIONALLY: no "does this chunk hold cells of the row" test in front of the
    // recurrence, the extension, the cut-off and the stores of every chunk, no record of when a chunk was last active, no NULL-back
    // stores (round 5: seven scalar instructions per test, ~55 of the 205 scalar instructions of a 128-diagonal step - on the unit
    // that bounds the kernel).  A row that leaves the chunks of its flavour, or fits a narrower one with the margin, ends the hot
    // loop; the frame code below shifts the rows and picks the flavour.
    constexpr int NA_MIN = NC >= 16 ? 4 : NC >= 8 ? 2 : 1;
    int na = NA_MIN;
    if (p.max_score < 1 || p.arena_cap < 1) status = 1;
    if (R16 && (plen > 12000 || tlen > 12000)) {
        status = 3;
        wide_at = W; // (never 0 with status 3: a 0 means 'not plain ACGT' to the host)
    }
    // the end test as ONE compare per chunk: tlen on the lane of the final diagonal, out of reach elsewhere
    int endk[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) endk[c] = kcol[c] == ak ? tlen : 2147483647;
    int s = 0;
    // this lane's cell (chunk 0) in the ring rows, by age: pM[a] = M[s-2a] (pM[0] is also where score 0 goes), pI / pD[0] = I / D[s],
    // [1] = I / D[s-2].  Rotated with the scores: five + four cheap register moves per score instead of the index arithmetic,
    // multiplications and address adds of "row = f(s mod 5)" on the scalar unit (which is what bounds the step).
    // (WR_LDS: the LDS address space on the device - 32-bit addresses and ds_* instructions however the pointers travel
    // through the loop; without it the rotated pointers are generic 64-bit ones and every access a flat_* instruction)
    typedef WR_LDS RT *LP;
    LP pM[5], pI[2], pD[2];
#pragma unroll
    // (based one cell below the lane's own: p[64 c] is diagonal k-1, p[64 c + 1] the cell itself, p[64 c + 2] diagonal k+1 -
    // all of them non-negative instruction offsets)
    for (int a = 0; a < 5; a++) pM[a] = (LP)ring + lane + ((5 - a) % 5) * RS;
    pI[0] = (LP)ring + lane + 5 * RS;
    pI[1] = (LP)ring + lane + 6 * RS;
    pD[0] = (LP)ring + lane + 7 * RS;
    pD[1] = (LP)ring + lane + 8 * RS;
    int32_t used = 1; // score 0 = one cell that is never read
    if (lane == 0) {
        p.hdr2[0] = 0;
        p.hdr2[1] = 0;
        p.hdr2[2] = 0;
        p.hdr2[3] = 1;
    }
    WR_WAVE_SYNC();
    bool bad = false;     // WIN: a byte that is not A / C / G / T was packed (the byte-comparing kernel takes the alignment)
    int qw0 = 0, tw0 = 0; // WIN: first resident word of either window (wave-uniform)
    if (WIN) l2_win_move2(qb, p.q, plen, &qw0, 0, tb, p.t, tlen, &tw0, 0, lane, &bad, true);
    // greedy extension of one cell per lane: h = offset of a valid cell on diagonal k (idle lanes: h = k = 0), 16 bases per pass
    // INTERIOR / EDGE.  Until some M cell has reached the end of either sequence (h = tlen or v = plen), every cell computed
    // from a valid source lies inside the DP matrix: an I / D / mismatch step adds at most one base on either axis, and an I or
    // D cell never leads the M cell of its own score and diagonal.  The trimmed ranges of the new wavefronts are then the
    // first / last cells that HAVE a valid source - scalar minima and maxima of the older ranges, no ballots, no bit scans, no
    // in-matrix compares.  That is all of an alignment but its last ~100 bases; `edge` is sticky from the first touch on.
    int edge_m = 0; // -1 from the first touch on (the hot loop tests signs)
    auto extend = [&](auto in_edge, bool valid, int h, int k) {
        const int hmax = tlen < plen + k ? tlen : plen + k;
        // a lane extends while h < lim; it stops by pulling lim down to h.  (The predicate is recomputed from registers every
        // pass: the wave mask of one compare is free, that of a loop-carried flag costs two vector instructions.)
        int lim = valid ? hmax : 0;
        // (guard + do-while: one compare + one conditional branch back per pass, nothing else of loop control - a while (true)
        // with a break in it came out of the compiler with two branches, a select and a mask test per pass inside the big kernel)
        if (WR_BALLOT(h < lim) != 0ull) {
            do { // (the positions of idle lanes stay inside the sequences)
                L2_COUNT(ext_pass, 1);
                const bool ext = h < lim;
                const uint32_t d = l2_get16(qb, h - k) ^ l2_get16(tb, h);
                const int nm = WR_CLZ(d) >> 1; // 16 when all 16 bases match
                h += ext ? nm : 0;
                lim = nm == 16 ? lim : h;
            } while (WR_BALLOT(h < lim) != 0ull);
        }
        // (an idle lane has h = 0 < hmax unless a sequence is empty - and then the EDGE copy, which is right anywhere, runs from the start)
        if (!decltype(in_edge)::value) edge_m = WR_UNIFORM(edge_m | (WR_BALLOT(h >= hmax) != 0ull ? -1 : 0));
        return h < hmax ? h : hmax;
    };
    // WIN: the cells of all chunks of a lane, 32 bases per pass through the windows.  A cell outside a window waits; once
    // nobody inside extends any more, both windows move to the smallest waiting positions (the cell with the smallest query
    // position is then inside both: two cells of a wavefront are less than W < 4000 diagonals apart) - k_wfa_lean's scheme.
    // h[c] / k[c]: offset and diagonal of a valid cell, 0 / 0 for an idle one; on[c]: the chunk has valid cells (wave-uniform).
    auto extend_win = [&](auto in_edge, auto na_tag, int *h, const int *k, const bool *valid) {
        constexpr int NA = decltype(na_tag)::value;
        int lim[NA], hmax[NA];
#pragma unroll
        for (int c = 0; c < NA; c++) {
            hmax[c] = tlen < plen + k[c] ? tlen : plen + k[c];
            lim[c] = valid[c] ? hmax[c] : 0;
        }
        while (true) {
            uint64_t pend = 0;
#pragma unroll
            for (int c = 0; c < NA; c++) {
                while (true) {
                    const uint64_t go_m = WR_BALLOT(h[c] < lim[c]) & WR_BALLOT(l2_win_has(qw0, h[c] - k[c])) & WR_BALLOT(l2_win_has(tw0, h[c]));
                    if (go_m == 0ull) break;
                    // (read by every lane: the slots are masked, any position is a valid LDS address)
                    uint32_t qh, ql, th, tl;
                    l2_win_get32(qb, h[c] - k[c], &qh, &ql);
                    l2_win_get32(tb, h[c], &th, &tl);
                    const uint32_t dh = qh ^ th, dl = ql ^ tl;
                    const int nm = dh ? WR_CLZ(dh) >> 1 : 16 + (WR_CLZ(dl) >> 1); // 32 when all 32 bases match
                    const bool go = h[c] < lim[c] && l2_win_has(qw0, h[c] - k[c]) && l2_win_has(tw0, h[c]);
                    h[c] += go ? nm : 0;
                    lim[c] = (!go || nm == 32) ? lim[c] : h[c];
                }
                pend |= WR_BALLOT(h[c] < lim[c]);
            }
            if (pend == 0ull) break;
            int mv = 2147483647, mh = 2147483647;
#pragma unroll
            for (int c = 0; c < NA; c++) {
                const bool wt = h[c] < lim[c];
                mh = wt && h[c] < mh ? h[c] : mh;
                mv = wt && h[c] - k[c] < mv ? h[c] - k[c] : mv;
            }
            mv = WR_WAVE_MIN_I32(mv);
            mh = WR_WAVE_MIN_I32(mh);
            l2_win_move2(qb, p.q, plen, &qw0, mv >> 4, tb, p.t, tlen, &tw0, mh >> 4, lane, &bad, false);
        }
#pragma unroll
        for (int c = 0; c < NA; c++) {
            if (!decltype(in_edge)::value) edge_m = WR_UNIFORM(edge_m | (WR_BALLOT(h[c] >= hmax[c]) != 0ull ? -1 : 0));
            h[c] = h[c] < hmax[c] ? h[c] : hmax[c];
        }
    };
    bool done = false;
    if (status == 0) { // score 0: the cell of diagonal 0 (chunk 0: its slot is 32 - ak / 2)
        const bool mine = kcol[0] == 0;
        int h;
        if (WIN) {
            int h_[1] = {0}, k_[1] = {0};
            bool v_[1] = {mine};
            extend_win(std::false_type{}, std::integral_constant<int, 1>{}, h_, k_, v_);
            h = h_[0];
        } else {
            h = extend(std::false_type{}, mine, 0, 0);
        }
        pM[0][1] = (RT)(mine ? h : RNULL);
        mlo[0] = mhi[0] = 0;
        done = ak == 0 && WR_READLANE(h, (0 - kbase) & 63) >= tlen;
        WR_WAVE_SYNC();
    }
    // Two loops.  The INNER one is the score step and nothing else: before a step it looks at the row the step would make
    // (from the ranges it already has) and leaves when anything but a plain step is due - the end, a limit, an empty row, a
    // row outside the chunks of the flavour or one that fits a narrower flavour, scratch running out.  The OUTER one does that
    // rare thing and comes back.  (With the rare paths inside the step, every variable they touch is merged on every path of
    // every step: a fifth of the step's instructions were register copies.)
    const int s_limit = R16 && p.max_score > 24000 ? 24000 : p.max_score; // (16-bit cells could wrap from s = 24000 on)
    int s_lim = s_limit; // the hot loop's copy: pulled below every score once the end is reached (one sign test covers both)
    int shrink_from = 0; // no "narrower flavour ?" test before this score (a live row may be wider than the new one for a while)
    int lo = 0, hi = 0; // the row of score s + 2 as the hot loop saw it when it left
    int32_t *hp = p.hdr2; // = p.hdr2 + s: carried along instead of recomputed from s (a 64-bit shift and add per score)
    // the hot loop, in 2 x (1 to 3) copies: INTERIOR (leaves also at the first touch) and EDGE, per flavour
    auto hot = [&](auto in_edge, auto na_tag) {
        constexpr bool EDGE = decltype(in_edge)::value;
        constexpr int NA = decltype(na_tag)::value;
        while (true) {
            // the row of score s + 2; sources: M[s-2] (mismatch), M[s-6] (gap open), I[s] / D[s] (gap extension)
            // (two-way minima pinned to the scalar unit: a three-way one is selected as v_min3_i32 + v_readfirstlane_b32)
            lo = WR_UNIFORM(mlo[1] < mlo[3] - 1 ? mlo[1] : mlo[3] - 1);
            hi = WR_UNIFORM(mhi[1] > mhi[3] + 1 ? mhi[1] : mhi[3] + 1);
            {
                const int l2 = WR_UNIFORM(ilo[0] + 1 < dlo[0] - 1 ? ilo[0] + 1 : dlo[0] - 1), h2 = WR_UNIFORM(ihi[0] + 1 > dhi[0] - 1 ? ihi[0] + 1 : dhi[0] - 1);
                lo = l2 < lo ? l2 : lo;
                hi = h2 > hi ? h2 : hi;
            }
            const uint32_t span = (uint32_t)(hi - lo); // (an empty row: far above W)
            // every "not a plain step" condition as the sign of one word (scalar adds and ORs, one compare): the end reached; the
            // score limit; an empty row (hi < lo); the row outside the chunks of the flavour; scratch; a narrower flavour would do
            uint32_t rare = (EDGE ? 0u : (uint32_t)edge_m) | (uint32_t)(s_lim - 3 - s) | span | (uint32_t)(lo - kbase) | (uint32_t)(kbase + 64 * NA - 1 - hi) |
                            ((uint32_t)p.arena_cap - (uint32_t)used - span - 1u);
            if (NA > NA_MIN) rare |= (uint32_t)((int)span + 2 * MARGIN - 32 * NA) & ~(uint32_t)(s + 2 - WR_UNIFORM(shrink_from));
            if ((int32_t)rare < 0) break;
            // ---- a plain step ----
            L2_COUNT(step[EDGE ? 1 : 0][NA > 4 ? 4 : NA], 1);
            L2_COUNT(width, (int)span + 1);
            s += 2;
            hp += 2;
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
                const LP t = pM[4]; // (the row of s-10: dead)
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
            // entry s/2 = {lo, row offset}; the offset of entry s/2+1 closes the row - the next step writes it as ITS row offset,
            // the last row's is written once after the loops.  EVERY lane stores the same two words to the same addresses (one
            // request): restricting it to lane 0 cost six scalar instructions of exec-mask traffic per score
            hp[0] = lo;
            hp[1] = rowb;
            const LP M8 = pM[4], M4 = pM[2], I2 = pI[1], D2 = pD[1]; // rows of s-8, s-4, s-2
            const int32_t rowk = WR_UNIFORM(rowb - lo); // byte of diagonal k: bt[rowk + k] (never negative for a cell of the row)
            int32_t off[NA], vins[NA], vdel[NA];
            // first / last cell inside the DP matrix of each of the three new wavefronts, as slots: lane order is diagonal order, so
            // these are ballots + ff1 / flbit on the scalar unit (none: first = 0xffffffff, last < 0)
            uint32_t fm = 0xffffffffu, fi = 0xffffffffu, fd = 0xffffffffu;
            int lm = -1, li = -1, ld = -1;
#pragma unroll
            for (int c = 0; c < NA; c++) {
                const int k = kcol[c];
                int32_t a = M8[64 * c], b = I2[64 * c];
                const bool iext = b >= a; // equal offsets: extension (lm_wfa_backtrace tags 2 > 1)
                const int32_t ins = (iext ? b : a) + 1;
                a = M8[64 * c + 2];
                b = D2[64 * c + 2];
                const bool dext = b >= a; // tags 4 > 3
                const int32_t del = dext ? b : a;
                const int32_t mis = (int32_t)M4[64 * c + 1] + 1;
                int32_t mx = mis > ins ? mis : ins;
                if (del > mx) mx = del;
                // predecessor of the M cell on equal offsets: mismatch (tag 9) > deletion (4, 3) > insertion (2, 1)
                const uint32_t mc = (mis >= del && mis >= ins) ? 0u : (del >= ins ? 2u : 1u);
                if ((uint32_t)(k - lo) <= span) p.bt[(uint32_t)(rowk + k)] = (uint8_t)(mc | (iext ? 4u : 0u) | (dext ? 8u : 0u));
                vins[c] = ins;
                vdel[c] = del;
                if (!EDGE) { // interior: a cell is valid when a source is (NULL + a few otherwise)
                    off[c] = mx < 0 ? RNULL : mx;
                    continue;
                }
                if ((uint32_t)mx > (uint32_t)tlen) mx = RNULL;
                if ((uint32_t)(mx - k) > (uint32_t)plen) mx = RNULL;
                off[c] = mx;
                // (cells outside [lo, hi] have NULL sources only, hence fail these tests by themselves; one ballot per compare: a
                // ballot of a conjunction costs two more vector instructions than the AND of two ballots)
                const uint64_t bm = WR_BALLOT(mx >= 0);
                const uint64_t bi = WR_BALLOT((uint32_t)ins <= (uint32_t)tlen) & WR_BALLOT((uint32_t)(ins - k) <= (uint32_t)plen);
                const uint64_t bd = WR_BALLOT((uint32_t)del <= (uint32_t)tlen) & WR_BALLOT((uint32_t)(del - k) <= (uint32_t)plen);
                // WR_FF1 / WR_FLB: s_ff1_i32_b64 / s_flbit_i32_b64, -1 for an empty mask: "| 64 c" keeps that above every slot,
                // "^ (64 c + 63)" turns the leading-zero count into the slot and an empty mask into a negative number
                const uint32_t f_m = (uint32_t)WR_FF1(bm) | (uint32_t)(64 * c), f_i = (uint32_t)WR_FF1(bi) | (uint32_t)(64 * c), f_d = (uint32_t)WR_FF1(bd) | (uint32_t)(64 * c);
                const int l_m = WR_FLB(bm) ^ (64 * c + 63), l_i = WR_FLB(bi) ^ (64 * c + 63), l_d = WR_FLB(bd) ^ (64 * c + 63);
                fm = f_m < fm ? f_m : fm;
                fi = f_i < fi ? f_i : fi;
                fd = f_d < fd ? f_d : fd;
                lm = l_m > lm ? l_m : lm;
                li = l_i > li ? l_i : li;
                ld = l_d > ld ? l_d : ld;
            }
            // (WR_UNIFORM: provably scalar - the ranges stay in scalar registers and so does everything derived from them)
            if (EDGE) {
                // (an empty range = anything beyond +-2^27: "none" is clamped to the sentinels and shifted by kbase like a slot - a
                // minimum and an add per end instead of compare, select, add)
                mlo[0] = WR_UNIFORM(kbase + (int)(fm < (uint32_t)E_LO ? fm : (uint32_t)E_LO));
                mhi[0] = WR_UNIFORM(kbase + (lm > E_HI ? lm : E_HI));
                ilo[0] = WR_UNIFORM(kbase + (int)(fi < (uint32_t)E_LO ? fi : (uint32_t)E_LO));
                ihi[0] = WR_UNIFORM(kbase + (li > E_HI ? li : E_HI));
                dlo[0] = WR_UNIFORM(kbase + (int)(fd < (uint32_t)E_LO ? fd : (uint32_t)E_LO));
                dhi[0] = WR_UNIFORM(kbase + (ld > E_HI ? ld : E_HI));
            } else {
                // interior: I[s] has a cell where M[s-8] or I[s-2] has one, one diagonal up; D[s] one diagonal down; M[s] where
                // M[s-4], I[s] or D[s] has one.  (Empty ranges are sentinels +- a few: they lose every minimum / maximum.)
                const int i_lo = WR_UNIFORM((mlo[4] < ilo[1] ? mlo[4] : ilo[1]) + 1), i_hi = WR_UNIFORM((mhi[4] > ihi[1] ? mhi[4] : ihi[1]) + 1);
                const int d_lo = WR_UNIFORM((mlo[4] < dlo[1] ? mlo[4] : dlo[1]) - 1), d_hi = WR_UNIFORM((mhi[4] > dhi[1] ? mhi[4] : dhi[1]) - 1);
                // (raw: an empty range is lo > hi here; the cut-off's clamps keep it empty and put the sentinels in, a row too
                // narrow for the cut-off does so itself below - one normalisation per step instead of two)
                ilo[0] = i_lo;
                ihi[0] = i_hi;
                dlo[0] = d_lo;
                dhi[0] = d_hi;
                // M[s] = the row: min(M[s-4].lo, I[s].lo, D[s].lo) is the `lo` of the top of this step term by term (never empty here)
                mlo[0] = lo;
                mhi[0] = hi;
                lm = 0; // "there is a valid M cell"
            }
            // ---- the new M cells, still in registers: greedy extension, end test, cut-off ----
            bool cut = false;
            if (lm >= 0) {
                if (WIN) {
                    int h_[NA], k_[NA];
                    bool v_[NA];
#pragma unroll
                    for (int c = 0; c < NA; c++) {
                        v_[c] = off[c] >= 0;
                        h_[c] = v_[c] ? off[c] : 0;
                        k_[c] = v_[c] ? kcol[c] : 0;
                    }
                    extend_win(in_edge, na_tag, h_, k_, v_);
#pragma unroll
                    for (int c = 0; c < NA; c++) off[c] = v_[c] ? h_[c] : RNULL;
                } else {
#pragma unroll
                    for (int c = 0; c < NA; c++) { // (a chunk without valid cells leaves at the first ballot)
                        const bool valid = off[c] >= 0;
                        const int h = extend(in_edge, valid, valid ? off[c] : 0, valid ? kcol[c] : 0);
                        off[c] = valid ? h : RNULL;
                    }
                }
                // the end: the cell of the final diagonal has reached the end of the target (the cut-off and the stores below
                // still run once: nothing reads them, and the step has no way out but its end).  endk is tlen on that lane only; an
                // invalid cell is NULL, below every tlen
                {
                    uint64_t eb = 0ull;
#pragma unroll
                    for (int c = 0; c < NA; c++) eb |= WR_BALLOT(off[c] >= endk[c]);
                    done = eb != 0ull;
                    s_lim = WR_UNIFORM(done ? -(1 << 30) : s_lim);
                }
                if (mhi[0] - mlo[0] + 1 >= 10) { // wf-adaptive(10, 50)
                    L2_COUNT(cutoff, 1);
                    int32_t dist[NA];
                    int32_t dm = 2147483647;
#pragma unroll
                    for (int c = 0; c < NA; c++) {
                        // (an invalid cell is NULL: its distance comes out beyond every valid one + 50 by itself)
                        const int32_t lv = plen - off[c] + kcol[c], lh = tlen - off[c];
                        dist[c] = lv > lh ? lv : lh;
                        dm = dist[c] < dm ? dist[c] : dm;
                    }
                    const int32_t dmin = WR_WAVE_MIN_I32(dm);
                    // The diagonals that stay: lm_wfa_align walks up from mlo to the first kept one below `top` and down from
                    // mhi to the last kept one above `bottom` = max(ak, new lo) - which is max(ak, mlo) (see k_wfa_lean).  With the
                    // lanes from `top` on counted as kept, the first kept lane IS the new lo (a kept cell below top if there is
                    // one, else top; never below mlo: kept cells are valid ones) - and the lanes up to `bottom` likewise for the new hi
                    const int top = ak < mhi[0] ? ak : mhi[0];
                    const int bottom = ak > mlo[0] ? ak : mlo[0];
                    uint32_t fl = 0xffffffffu;
                    int lh_ = -1;
#pragma unroll
                    for (int c = 0; c < NA; c++) {
                        const uint64_t keep = WR_BALLOT(dist[c] - dmin <= 50);
                        const uint64_t kl = keep | WR_BALLOT(kcol[c] >= top), kh = keep | WR_BALLOT(kcol[c] <= bottom);
                        const uint32_t f = (uint32_t)WR_FF1(kl) | (uint32_t)(64 * c);
                        const int l = WR_FLB(kh) ^ (64 * c + 63);
                        fl = f < fl ? f : fl;
                        lh_ = l > lh_ ? l : lh_;
                    }
                    int nlo = kbase + (int)fl, nhi = kbase + lh_; // (top and bottom are inside the chunks of the flavour: both exist)
                    nlo = nlo > mlo[0] ? nlo : mlo[0];
                    nhi = nhi < mhi[0] ? nhi : mhi[0];
                    // I[s] / D[s] are clamped to the reduced M range (empty stays empty: the sentinels survive max / min).  INTERIOR:
                    // their ranges lie inside M's (a cell with a valid I or D source has a valid M), so without a cut the clamps
                    // change nothing - applied without asking (ten scalar instructions; the test and its branch were eight, the
                    // clamps fourteen more on most steps of a wavefront in steady state).  EDGE: an M cell may be outside the
                    // matrix where its I / D cell is inside, so the ranges are only touched by a real cut.
                    if (!EDGE || nlo != mlo[0] || nhi != mhi[0]) {
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
                } else if (!EDGE) { // (no cut-off on a row of fewer than ten cells: the interior ranges get their sentinels here)
                    if (ilo[0] > ihi[0]) {
                        ilo[0] = E_LO;
                        ihi[0] = E_HI;
                    }
                    if (dlo[0] > dhi[0]) {
                        dlo[0] = E_LO;
                        dhi[0] = E_HI;
                    }
                }
            }
            // ---- the three rows of score s: cells outside a range are NULL ----
            {
                const uint32_t spm = (uint32_t)(mhi[0] - mlo[0]), spi = (uint32_t)(ihi[0] - ilo[0]), spd = (uint32_t)(dhi[0] - dlo[0]);
#pragma unroll
                for (int c = 0; c < NA; c++) { // (k - E_LO) as unsigned is above every span, also above the span of an empty range
                    const int k = kcol[c];
                    int32_t m = off[c];
                    if (cut) m = (uint32_t)(k - mlo[0]) <= spm ? m : RNULL; // (without a cut the cells outside the range are NULL already)
                    newM[64 * c + 1] = (RT)m;
                    newI[64 * c + 1] = (RT)((uint32_t)(k - ilo[0]) <= spi ? vins[c] : RNULL);
                    newD[64 * c + 1] = (RT)((uint32_t)(k - dlo[0]) <= spd ? vdel[c] : RNULL);
                }
            }
            WR_WAVE_SYNC(); // the rows of score s are in the ring
        }
    };
    // the flavours this ring width has: NA_MIN, then twice and four times that while they fit
    constexpr int F0 = NA_MIN, F1 = 2 * F0 <= NC ? 2 * F0 : NC, F2 = 4 * F0 <= NC ? 4 * F0 : NC;
    auto run_hot = [&](auto in_edge) {
        if (na == F0)
            hot(in_edge, std::integral_constant<int, F0>{});
        else if (F1 != F0 && na == F1)
            hot(in_edge, std::integral_constant<int, F1>{});
        else
            hot(in_edge, std::integral_constant<int, F2>{});
    };
    while (status == 0 && !done) {
        if (edge_m)
            run_hot(std::true_type{});
        else
            run_hot(std::false_type{});
        // ---- what is due instead of a plain step (lo, hi: the row of score s + 2; possibly nothing but the first touch) ----
        if (done) break;
        if (s + 2 >= s_limit) {
            status = s + 2 >= p.max_score ? 1 : 3;
            wide_at = W;
            break;
        }
        if (lo > hi) { // no source wavefront (all four empty): an empty row
            s += 2;
            hp += 2;
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
            for (int c = 0; c < NC; c++) pM[0][64 * c + 1] = pI[0][64 * c + 1] = pD[0][64 * c + 1] = (RT)RNULL;
            hp[0] = 0; // same offset as the next row
            hp[1] = used;
            WR_WAVE_SYNC();
            continue;
        }
        if ((int64_t)used + (hi - lo + 1) > (int64_t)p.arena_cap) {
            status = 1;
            break;
        }
        // The frame.  Every live row - M[s] .. M[s-6], I / D[s] (I may start at lo - 1, D may end at hi + 1) - and the new one
        // must be inside the chunks of the flavour; the flavour is the narrowest that holds them with the margin.
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
            int f = NA_MIN; // the flavour the live rows get (NC when the margin does not fit anywhere: uw <= W)
            while (f < NC && 64 * f < uw + 2 * MARGIN) f *= 2;
            const bool out = lo < kbase || hi > kbase + 64 * na - 1;
            if (!out && f >= na) { // the new row asks for a narrower flavour, the live rows do not allow it yet (or: the first touch)
                shrink_from = WR_UNIFORM(s + 2 + 8);
                continue;
            }
            // shift the nine rows: the live rows centred on the chunks of the flavour (every cell of every row is rewritten:
            // what lies outside the live rows becomes NULL)
            const int nk = WR_UNIFORM(ulo - (64 * f - uw) / 2);
            const int delta = nk - kbase; // new slot i <- old slot i + delta
#pragma unroll 1
            for (int r = 0; r < 9; r++) {
                RT v[NC];
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    const int src = lane + 64 * c + delta;
                    v[c] = (uint32_t)src < (uint32_t)W ? ring[r * RS + 1 + src] : (RT)RNULL;
                }
                WR_WAVE_SYNC(); // every lane has read the row
#pragma unroll
                for (int c = 0; c < NC; c++) cell0[r * RS + 64 * c] = v[c];
            }
            WR_WAVE_SYNC();
            kbase = nk;
            na = WR_UNIFORM(f);
            shrink_from = WR_UNIFORM(s + 2 + 8);
#pragma unroll
            for (int c = 0; c < NC; c++) {
                kcol[c] = nk + lane + 64 * c;
                endk[c] = kcol[c] == ak ? tlen : 2147483647;
            }
            nrec++;
        }
    }
    hp[3] = used; // the last row's closing offset (score 0: written above)
    if (WIN && status == 0 && WR_BALLOT(bad) != 0ull) status = 3; // not plain ACGT: the result is discarded
    res->qw0 = qw0;
    res->tw0 = tw0;
    res->status = status;
    res->score = status == 0 ? s : wide_at;
    res->used = used;
    res->recentres = nrec;
}
