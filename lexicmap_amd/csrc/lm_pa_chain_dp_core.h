// pa_chain_dp.h - the banded DP of Chainer2 (lib-chaining2.go:222-307; k_pa_chain_wave, lm_kernels.hip) for ONE wavefront per
// chain, with the recent anchors and their scores in an LDS ring.
//
// Why: anchor i depends on the scores of the up to ~50 anchors before it, so the loop over i is serial, and in
// k_pa_chain_wave every step goes through global memory: the score of anchor i-1 is stored by lane 0, the workgroup barrier
// waits for the store, the next step loads anchors and scores back (~1.5 us per anchor: 150 ms for one 10^5-anchor window of
// a C4 query, 0.6 s of its 1.86-s step; profiles/r03_c4_shard0_of_4.json).  Here the last PCD_RING anchors {qbegin, tbegin,
// len, score} live in LDS, the next 64 anchors are fetched together, the 64-bit (score, ~j) maximum is a DPP reduction, and
// nothing in the step waits for global memory; candidates further back than the ring (possible only while the band holds
// more than PCD_RING anchors) are read from global memory as before.  Scores, predecessors, the best score and its anchor are
// identical to lm_run_chain2's.  One source for the device and for the host emulator (simt_emu.h).
#pragma once
#include <stdint.h>

#define PCD_RING 128 /* anchors held in LDS: two candidate rounds of 64 */

struct PcdLds {
    int32_t q[PCD_RING], t[PCD_RING], len[PCD_RING];
    uint32_t score[PCD_RING];
    int32_t nq[64], nt[64], nlen[64]; // the next 64 anchors
};

// a_[0..n): the cleared + trimmed anchors (n >= 2); msi[i] = score << 32 | predecessor, as lm_run_chain2 leaves it.  Returns the
// best score in *M and its anchor in *Mi (identical in all lanes).
// pa_chain_dp_reg - the DP with the last 64 anchors and their scores in REGISTERS: lane l holds anchor i - 1 - l, so the first
// candidate round of anchor i (in nearly every window the only one: the band closes after 50 candidates or 100 bases) reads
// nothing from LDS, and its result - the largest score, the farthest of the candidates that reach it - is a 32-bit wave maximum,
// one ballot and a bit scan instead of the 64-bit key reduction; the finished anchor enters at lane 0 by a one-lane wavefront
// shift of the four registers (DPP wave_shr:1).  Its predecessor (round 4: every candidate round out of the LDS ring) was a chain of five
// LDS round trips and twelve 64-bit DPP steps per anchor (~1 us whatever the chip does beside it; C4 shard 1.45 -> 1.33 s per
// step with this form, k_pa_chain 1.23 -> 0.89 s of exclusive time per C3 step); this one has none on its critical path.  The
// LDS ring is still written (one store per anchor) for the rare further rounds.
// The includer adds PCD_SHIFT_IN(newv, v) (lane 0 <- the wave-uniform newv, lane l <- lane l - 1's v) and PCD_WAVE_MAX_I32(v).
PCD_DEV void pa_chain_dp_reg(const LmSub *a_, int n, const LmChain2Opt &opt, uint64_t *msi, PcdLds *L, long long *Mout, int *Miout) {
    const int lane = PCD_LANE;
    const unsigned long long le_mask = (lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull);
    long long M = 0;
    int Mi = 0;
    int32_t rq = 0, rt = 0, rlen = 0; // anchor i - 1 - lane (valid while lane < i)
    uint32_t rs = 0;                  // its score
    for (int i0 = 0; i0 < n; i0 += 64) {
        PCD_LDS_SYNC(); // everybody is done with the previous 64 anchors
        if (i0 + lane < n) {
            const LmSub x = a_[i0 + lane];
            L->nq[lane] = x.qbegin;
            L->nt[lane] = x.tbegin;
            L->nlen[lane] = (int32_t)x.len;
        }
        PCD_LDS_SYNC();
        const int i1 = i0 + 64 < n ? i0 + 64 : n;
        for (int i = i0; i < i1; i++) {
            const int32_t aq = L->nq[i - i0], at = L->nt[i - i0], alen = L->nlen[i - i0];
            long long m = alen;
            int mj = i;
            if (i > 0) {
                // ---- the first round out of the registers ----
                const bool skip = lane >= i || rq == aq || rt > at;
                const unsigned long long nskip = PCD_BALLOT(!skip);
                const int cnt = PCD_POPCLL(nskip & le_mask);
                bool brk = false;
                if (!skip) {
                    const int32_t bbase = aq - rq - rlen;
                    brk = !(bbase <= opt.band_base || cnt <= opt.band_count);
                }
                const unsigned long long bm = PCD_BALLOT(brk);
                const int first_brk = bm ? (PCD_FFSLL(bm) - 1) : 64;
                int32_t sc = -1;
                if (!skip && lane < first_brk) {
                    int32_t qd = aq - rq, td = at - rt;
                    if (qd < 0) qd = -qd;
                    if (td < 0) td = -td;
                    const int32_t g = qd > td ? qd - td : td - qd;
                    if (g <= opt.max_gap) {
                        const long long s = (long long)rs + (long long)rlen - (long long)g;
                        if (s >= 0) sc = (int32_t)s; // (scores are sums of at most n anchor lengths: far below 2^31)
                    }
                }
                if (bm != 0ull || i <= 64) { // the band closed, or there is nothing further back: the usual case
                    const int32_t mx = PCD_WAVE_MAX_I32(sc);
                    if (mx >= 0 && (long long)mx >= m) {
                        const unsigned long long eq = PCD_BALLOT(sc == mx);
                        m = mx;
                        mj = i - 1 - (63 - PCD_CLZLL(eq)); // the farthest candidate with that score (the key's tie rule)
                    }
                } else {
                    // ---- further rounds from the LDS ring / global memory ----
                    unsigned long long best = sc >= 0 ? (((unsigned long long)(uint32_t)sc << 32) | (unsigned long long)(0xffffffffu - (uint32_t)(i - 1 - lane))) : 0ull;
                    int bcount = PCD_POPCLL(nskip);
                    bool stop = false;
                    PCD_LDS_SYNC(); // lane 0's ring stores of the earlier anchors are visible
                    for (int jt = i - 65; jt >= 0 && !stop; jt -= 64) {
                        const int j = jt - lane;
                        const bool inb = j >= 0;
                        int32_t bq = 0, bt = 0, blen = 0;
                        uint32_t bs = 0;
                        if (i - (jt - 63) <= PCD_RING) { // the whole round is in the ring (uniform)
                            const int sl = j & (PCD_RING - 1);
                            bq = L->q[sl];
                            bt = L->t[sl];
                            blen = L->len[sl];
                            bs = L->score[sl];
                        } else {
                            PCD_GLOBAL_FENCE(); // lane 0's stores of the scores are visible to the loads below
                            if (inb) {
                                const LmSub b = a_[j];
                                bq = b.qbegin;
                                bt = b.tbegin;
                                blen = (int32_t)b.len;
                                bs = (uint32_t)(msi[j] >> 32);
                            }
                        }
                        const bool skip2 = !inb || bq == aq || bt > at;
                        const unsigned long long nskip2 = PCD_BALLOT(!skip2);
                        const int cnt2 = bcount + PCD_POPCLL(nskip2 & le_mask);
                        bool brk2 = false;
                        if (!skip2) {
                            const int32_t bbase = aq - bq - blen;
                            brk2 = !(bbase <= opt.band_base || cnt2 <= opt.band_count);
                        }
                        const unsigned long long bm2 = PCD_BALLOT(brk2);
                        const int fb2 = bm2 ? (PCD_FFSLL(bm2) - 1) : 64;
                        if (bm2) stop = true;
                        if (!skip2 && lane < fb2) {
                            int32_t qd = aq - bq, td = at - bt;
                            if (qd < 0) qd = -qd;
                            if (td < 0) td = -td;
                            const int32_t g = qd > td ? qd - td : td - qd;
                            if (g <= opt.max_gap) {
                                const long long s = (long long)bs + (long long)blen - (long long)g;
                                if (s >= 0) {
                                    const unsigned long long key = ((unsigned long long)s << 32) | (unsigned long long)(0xffffffffu - (uint32_t)j);
                                    if (key > best) best = key;
                                }
                            }
                        }
                        bcount += PCD_POPCLL(nskip2);
                    }
                    best = PCD_WAVE_MAX_U64(best);
                    if (best != 0) {
                        const long long s = (long long)(best >> 32);
                        if (s >= m) {
                            m = s;
                            mj = (int)(0xffffffffu - (uint32_t)(best & 0xffffffffu));
                        }
                    }
                }
            }
            if (lane == 0) { // (LDS operations of one wavefront complete in order: the ring slot's last readers are done)
                msi[i] = ((uint64_t)m << 32) | (uint64_t)(uint32_t)mj;
                const int sl = i & (PCD_RING - 1);
                L->q[sl] = aq;
                L->t[sl] = at;
                L->len[sl] = alen;
                L->score[sl] = (uint32_t)m;
            }
            rq = PCD_SHIFT_IN(aq, rq);
            rt = PCD_SHIFT_IN(at, rt);
            rlen = PCD_SHIFT_IN(alen, rlen);
            rs = (uint32_t)PCD_SHIFT_IN((int32_t)(uint32_t)m, (int32_t)rs);
            if (i > 0 && m > M) { // (the best score is sought among anchors 1.., as in lm_run_chain2)
                M = m;
                Mi = i;
            }
        }
    }
    *Mout = M;
    *Miout = Mi;
}
