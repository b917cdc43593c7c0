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

// a_[0..n): the cleared + trimmed anchors (n >= 2); msi[i] = score << 32 | predecessor, as lm_run_chain2 leaves it.
// Returns the best score in *M and its anchor in *Mi (identical in all lanes).
PCD_DEV void pa_chain_dp_ring(const LmSub *a_, int n, const LmChain2Opt &opt, uint64_t *msi, PcdLds *L, long long *Mout, int *Miout) {
    const int lane = PCD_LANE;
    long long M = 0;
    int Mi = 0;
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
                unsigned long long best = 0; // (score << 32 | ~j) of the best candidate so far, 0 = none
                int bcount = 0;
                bool stop = false;
                for (int jt = i - 1; jt >= 0 && !stop; jt -= 64) {
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
                    const bool skip = !inb || bq == aq || bt > at;
                    const unsigned long long nskip = PCD_BALLOT(!skip);
                    const int cnt = bcount + PCD_POPCLL(nskip & ((lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull)));
                    bool brk = false;
                    if (!skip) {
                        const int32_t bbase = aq - bq - blen;
                        brk = !(bbase <= opt.band_base || cnt <= opt.band_count);
                    }
                    const unsigned long long bm = PCD_BALLOT(brk);
                    const int first_brk = bm ? (PCD_FFSLL(bm) - 1) : 64;
                    if (bm) stop = true;
                    if (!skip && lane < first_brk) {
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
                    bcount += PCD_POPCLL(nskip);
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
            PCD_LDS_SYNC(); // every lane has read the ring slot that is about to be overwritten (anchor i - PCD_RING)
            if (lane == 0) {
                msi[i] = ((uint64_t)m << 32) | (uint64_t)(uint32_t)mj;
                const int sl = i & (PCD_RING - 1);
                L->q[sl] = aq;
                L->t[sl] = at;
                L->len[sl] = alen;
                L->score[sl] = (uint32_t)m;
            }
            PCD_LDS_SYNC();
            if (i > 0 && m > M) { // (the best score is sought among anchors 1.., as in lm_run_chain2)
                M = m;
                Mi = i;
            }
        }
    }
    *Mout = M;
    *Miout = Mi;
}
