// lm_pa_chain_pipe_dp.h - STAGED for round 5 (not in lexicmap_amd/csrc): the banded DP of Chainer2
// (lib-chaining2.go:222-307; k_pa_chain_wave, lm_kernels.hip) by a WORKGROUP of PCP_NW wavefronts per chain window, as a
// pipeline over the anchors.
//
// Why: anchor i depends on the scores of the ~50 anchors before it, so k_pa_chain_wave walks a window one anchor at a time:
// ~185 dependent instructions of ONE wavefront per anchor (64-wide candidate scan, two ballots, a 64-bit DPP reduction), ~1 us
// each whatever the memory does (the LDS ring of round 4 removed the global round trip: 152 -> 130 ms per C4 launch, no more).
// A 10^5-anchor window of a C4 query is 100 ms of one wavefront while the chip idles: 0.52 s of the 1.65-s C4 shard step,
// 0.8 s of the 12.3-s C3 step.  But everything except the last addition is independent of the scores: which candidates
// j < i are in anchor i's band (skip rule, count, band break, gap) depends on COORDINATES only.  So wavefront w takes anchors
// w, w + NW, w + 2 NW, ...: for its anchor it evaluates the band and the additive term of all 64 candidates ahead of time,
// reduces the candidates whose scores are final already (the wavefronts behind it in the pipeline have published them) to a
// partial maximum, and then takes the few still-pending candidates (at most ~2 NW) one by one in order, as their scores
// appear.  The serial chain per anchor shrinks from the whole step to: see the predecessor's score (one LDS read), one
// add / compare, publish (one LDS write + the counter).  Scores, predecessors, the best score and its anchor are those of
// lm_run_chain2 (lm_algos.h); checked on the host SIMT emulator (tests/test_pa_chain_pipe_emulated_cpu.py) against it.
//
// One source for the device and the emulator: the includer defines
//   PCP_DEV, PCP_TID, PCP_BALLOT(p), PCP_WAVE_SYNC(), PCP_BARRIER(), PCP_POPCLL, PCP_FFSLL, PCP_WAVE_MAX_U64(v),
//   PCP_BCAST32(v, lane)  value of `lane` of the caller's wavefront (all 64 lanes call: v_readlane / shuffle),
//   PCP_LOAD_DONE(p)      a wave-UNIFORM volatile read of the progress counter (lane 0 reads, all lanes get it),
//   PCP_STORE_DONE(p, v)  publish (after the score is written: release order), PCP_SPIN() one polite spin step (s_sleep),
//   PCP_GLOBAL_FENCE()    stores of other wavefronts to msi[] are visible after it,
//   PCP_LOAD_MSI(p)       a load of msi[] that does not come from a stale first-level cache line (device: glc / relaxed atomic).
#pragma once
#include <stdint.h>

#ifndef PCP_SCHED_POINT
#define PCP_SCHED_POINT() /* the emulator perturbs the interleaving of the wavefronts here */
#endif
#ifndef PCP_NW
#define PCP_NW 8 /* wavefronts per window */
#endif
#define PCP_RING 256 /* scores kept in LDS: >= 64 (one candidate round) + the pipeline's depth */

struct PcpLds {
    uint32_t score[PCP_RING];
    volatile int done; // anchors 0 .. done-1 have their final score in score[] / msi[]
    long long best[PCP_NW];
    int best_i[PCP_NW];
};

// a_[0..n): the cleared + trimmed anchors (n >= 2); msi[i] = score << 32 | predecessor as lm_run_chain2 leaves it.
// All PCP_NW * 64 threads of the workgroup call; *Mout / *Miout are the same in every thread afterwards.
PCP_DEV void pa_chain_dp_pipe(const LmSub *a_, int n, const LmChain2Opt &opt, uint64_t *msi, PcpLds *L, long long *Mout, int *Miout) {
    const int tid = PCP_TID, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) L->done = 0;
    PCP_BARRIER();
    long long M = 0;
    int Mi = 0;
    for (int i = wave; i < n; i += PCP_NW) {
        const LmSub a = a_[i];
        const int32_t aq = a.qbegin, at = a.tbegin, alen = (int32_t)a.len;
        long long m = alen;
        int mj = i;
        if (i > 0) {
            unsigned long long best = 0; // (score << 32 | ~j) of the best candidate so far, 0 = none
            int bcount = 0;
            bool stop = false;
            for (int jt = i - 1; jt >= 0 && !stop; jt -= 64) {
                const int j = jt - lane;
                const bool inb = j >= 0;
                // ---- what depends on coordinates only: band membership and the additive term of this lane's candidate ----
                int32_t bq = 0, bt = 0, blen = 0;
                if (inb) {
                    const LmSub b = a_[j];
                    bq = b.qbegin;
                    bt = b.tbegin;
                    blen = (int32_t)b.len;
                }
                const bool skip = !inb || bq == aq || bt > at;
                const unsigned long long nskip = PCP_BALLOT(!skip);
                const int cnt = bcount + PCP_POPCLL(nskip & ((lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull)));
                bool brk = false;
                if (!skip) {
                    const int32_t bbase = aq - bq - blen;
                    brk = !(bbase <= opt.band_base || cnt <= opt.band_count);
                }
                const unsigned long long bm = PCP_BALLOT(brk);
                const int first_brk = bm ? (PCP_FFSLL(bm) - 1) : 64;
                if (bm) stop = true;
                bool valid = false;
                int32_t add = 0;
                if (!skip && lane < first_brk) {
                    int32_t qd = aq - bq, td = at - bt;
                    if (qd < 0) qd = -qd;
                    if (td < 0) td = -td;
                    const int32_t g = qd > td ? qd - td : td - qd;
                    valid = g <= opt.max_gap;
                    add = blen - g;
                }
                bcount += PCP_POPCLL(nskip);
                PCP_SCHED_POINT();
                // ---- the scores: final ones in bulk, pending ones in order as they appear ----
                const int jlo = jt - 63 > 0 ? jt - 63 : 0; // candidates of this round: jlo .. jt
                int d = PCP_LOAD_DONE(&L->done);           // wave-uniform snapshot
                if (jt != i - 1) {                         // an older round: every candidate is at least 64 behind - wait for all
                    while (d <= jt) {
                        PCP_SPIN();
                        d = PCP_LOAD_DONE(&L->done);
                    }
                }
                const bool in_ring = i - jlo <= PCP_RING - 2 * PCP_NW; // nobody can have overwritten these slots yet
                if (!in_ring) PCP_GLOBAL_FENCE();
                if (valid && j < d) {
                    const uint32_t bs = in_ring ? L->score[j & (PCP_RING - 1)] : (uint32_t)(PCP_LOAD_MSI(&msi[j]) >> 32);
                    const long long s = (long long)bs + (long long)add;
                    if (s >= 0) {
                        const unsigned long long key = ((unsigned long long)s << 32) | (unsigned long long)(0xffffffffu - (uint32_t)j);
                        if (key > best) best = key;
                    }
                }
                // pending: j = max(d, jlo) .. jt (only in the first round; few: the pipeline is PCP_NW deep), ascending
                for (int jp = d > jlo ? d : jlo; jp <= jt; jp++) {
                    while (PCP_LOAD_DONE(&L->done) <= jp) PCP_SPIN();
                    const int src = jt - jp; // the lane that holds candidate jp
                    const uint32_t v_ok = PCP_BCAST32(valid ? 1u : 0u, src);
                    const uint32_t v_add = PCP_BCAST32((uint32_t)add, src);
                    if (v_ok) {
                        const long long s = (long long)L->score[jp & (PCP_RING - 1)] + (long long)(int32_t)v_add;
                        if (s >= 0) {
                            const unsigned long long key = ((unsigned long long)s << 32) | (unsigned long long)(0xffffffffu - (uint32_t)jp);
                            if (lane == 0 && key > best) best = key; // (kept in one lane: the reduction below takes the maximum)
                        }
                    }
                }
            }
            best = PCP_WAVE_MAX_U64(best);
            if (best != 0) {
                const long long s = (long long)(best >> 32);
                if (s >= m) {
                    m = s;
                    mj = (int)(0xffffffffu - (uint32_t)(best & 0xffffffffu));
                }
            }
        }
        // ---- publish, in anchor order: anchor i - 1 is final (we read it above, or i == 0) ----
        PCP_SCHED_POINT();
        while (PCP_LOAD_DONE(&L->done) < i) PCP_SPIN(); // (i == 0, or a wavefront whose band was empty, gets here without having waited)
        if (lane == 0) {
            msi[i] = ((uint64_t)m << 32) | (uint64_t)(uint32_t)mj;
            L->score[i & (PCP_RING - 1)] = (uint32_t)m;
        }
        PCP_WAVE_SYNC();
        if (lane == 0) PCP_STORE_DONE(&L->done, i + 1);
        PCP_SCHED_POINT();
        if (i > 0 && m > M) { // (the best score is sought among anchors 1.., as in lm_run_chain2; ties: the smallest anchor)
            M = m;
            Mi = i;
        }
    }
    if (lane == 0) {
        L->best[wave] = M;
        L->best_i[wave] = Mi;
    }
    PCP_BARRIER();
    long long Mb = 0;
    int Mib = 0;
    for (int w = 0; w < PCP_NW; w++) {
        const long long mw = L->best[w];
        const int iw = L->best_i[w];
        if (mw > Mb || (mw == Mb && mw > 0 && iw < Mib)) {
            Mb = mw;
            Mib = iw;
        }
    }
    PCP_BARRIER();
    *Mout = Mb;
    *Miout = Mib;
}
