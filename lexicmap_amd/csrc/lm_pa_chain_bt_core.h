// lm_pa_chain_bt_core.h - the backtrack of Chainer2 (lm_run_chain2's second half; lib-chaining2.go:309-420 as
// restated in lm_algos.h) by a WAVEFRONT instead of one lane.  Equal to lm_run_chain2 on the host SIMT
// emulator (tests/test_pa_chain_bt_emulated_cpu.py) and, in k_pa_chain_wave, to the oracle's rows on the GPU (round 5).
//
// Why: k_pa_chain_wave runs the backtrack on lane 0: a walk from anchor to predecessor with two dependent GLOBAL loads per
// anchor (msi[i], then the anchor), and, for every region left and right of a chain, a serial scan of the region for its best
// score.  For a window of n anchors that is n round trips of ~0.5 us by one lane plus O(n) serial loads per region - by the
// arithmetic of its loads as long as ClearSubstrPairs and ten times the DP; on C4's 10 000-anchor windows,
// milliseconds per window, which is what the 130-ms launches wait for (the LDS ring of round 4 took the global round trip out
// of the DP and gained only 15 %).  Here:
//  * the region scan is the wavefront's: 64 lanes stride over the region, one 64-bit maximum (score << 32 | ~index: the first
//    best anchor, as the serial scan's strict ">" picks it);
//  * the walk reads (score, predecessor, anchor) from a 64-anchor tile in LDS that the wavefront loads with one coalesced round
//    trip when the walk leaves it - predecessors are a few anchors back, so a tile serves tens of steps;
//  * everything the walk computes is wave-uniform (LDS broadcast reads): all lanes run it, lane 0 owns the region stack and the
//    output records.
// Same results, bit for bit (the double arithmetic of pident included), in the same order.
#pragma once
#include <stdint.h>

#define PCB_TILE 64
#ifndef PCB_COUNT
#define PCB_COUNT(what) /* the emulator harness counts walk steps and tile loads */
#endif
struct PcbLds {
    uint64_t msi[PCB_TILE];
    int32_t q[PCB_TILE], t[PCB_TILE], len[PCB_TILE];
};

// a_[0..n): cleared + trimmed anchors (n >= 2), msi[i] = score << 32 | predecessor (the DP's result), M / Mi its best score and
// anchor; stack: 2 n + 4 ints of global scratch owned by lane 0; res: room for n records.  Returns the number of chains
// (identical in all lanes); res[] is written by lane 0.  All 64 lanes call.
PCB_DEV int pa_chain_backtrack_wave(const LmSub *a_, int n, const LmChain2Opt &opt, const uint64_t *msi, long long M, int Mi, int32_t *stack,
                                    LmChain2 *res, PcbLds *L) {
    const int lane = PCB_LANE;
    int nout = 0;
    if (M < (long long)opt.min_score) return 0;
    int sp = 0; // (wave-uniform; the entries themselves are lane 0's)
    if (lane == 0) {
        stack[0] = 0;
        stack[1] = n;
    }
    sp = 2;
    int pending_Mi0 = Mi;
    int tb = -(1 << 30); // first anchor of the tile in LDS (none yet)
    while (sp > 0) {
        int hi = 0, lo = 0;
        if (lane == 0) {
            hi = stack[sp - 1];
            lo = stack[sp - 2];
        }
        sp -= 2;
        hi = PCB_UNIFORM(hi);
        lo = PCB_UNIFORM(lo);
        int mi;
        if (pending_Mi0 >= 0) {
            mi = pending_Mi0;
            pending_Mi0 = -1;
        } else { // the best anchor of the region: the first one with the highest score
            unsigned long long best = 0;
            for (int i = lo + lane; i < hi; i += 64) {
                const unsigned long long key = (msi[i] & 0xffffffff00000000ull) | (unsigned long long)(0xffffffffu - (uint32_t)i);
                best = key > best ? key : best;
            }
            best = PCB_WAVE_MAX_U64(best);
            const long long bestm = (long long)(best >> 32);
            mi = bestm > 0 ? (int)(0xffffffffu - (uint32_t)(best & 0xffffffffull)) : lo;
            if (bestm < (long long)opt.min_score) continue;
        }
        int n_matched = 0, n_abq = 0, n_abt = 0;
        int i = mi, j = 0;
        int32_t qb = 0, qe = 0, tbg = 0, te = 0;
        int begin_of_next = 0;
        bool first_anchor = true, jneg = false;
        int n_anchors = 0;
        while (true) {
            if ((uint32_t)(i - tb) >= (uint32_t)PCB_TILE) { // the walk left the tile: the 64 anchors ending at i
                PCB_LDS_SYNC(); // every lane is done with the old tile
                PCB_COUNT(1);
                tb = i - (PCB_TILE - 1) > 0 ? i - (PCB_TILE - 1) : 0;
                const int g = tb + lane;
                if (g < n) {
                    const LmSub s = a_[g];
                    L->msi[lane] = msi[g];
                    L->q[lane] = s.qbegin;
                    L->t[lane] = s.tbegin;
                    L->len[lane] = (int32_t)s.len;
                }
                PCB_LDS_SYNC();
            }
            const int o = i - tb;
            PCB_COUNT(0);
            j = (int)(L->msi[o] & 4294967295ull);
            if (j < lo) {
                jneg = true;
                break;
            }
            const int32_t sq = L->q[o], st = L->t[o], sl = L->len[o];
            n_anchors++;
            if (first_anchor) {
                first_anchor = false;
                qe = sq + sl - 1;
                te = st + sl - 1;
                qb = sq;
                tbg = st;
                n_matched += sl;
            } else {
                qb = sq;
                tbg = st;
                if (sq + sl - 1 >= begin_of_next)
                    n_matched += begin_of_next - sq;
                else
                    n_matched += sl;
            }
            begin_of_next = sq;
            if (i == j) {
                n_abq += (int)qe - (int)qb + 1;
                if (n_abq < opt.min_align_len) break;
                n_abt += (int)te - (int)tbg + 1;
                double pident = (double)n_matched / (double)(n_abq > n_abt ? n_abq : n_abt) * 100;
                if (pident < opt.heuristic_pident) break;
                if (pident > 100) pident = 100;
                if (lane == 0) {
                    LmChain2 p;
                    p.nanchors = n_anchors;
                    p.aligned_bases_q = n_abq;
                    p.aligned_bases_t = n_abt;
                    p.matched_bases = n_matched;
                    p.pident = pident;
                    p.qbegin = qb;
                    p.qend = qe;
                    p.tbegin = tbg;
                    p.tend = te;
                    res[nout] = p;
                }
                nout++;
                break;
            }
            i = j;
        }
        if (jneg && n_anchors > 0) {
            n_abq += (int)qe - (int)qb + 1;
            n_abt += (int)te - (int)tbg + 1;
            if (n_abq >= opt.min_align_len) {
                double pident = (double)n_matched / (double)(n_abq > n_abt ? n_abq : n_abt) * 100;
                if (pident >= opt.heuristic_pident) {
                    if (pident > 100) pident = 100;
                    if (lane == 0) {
                        LmChain2 p;
                        p.nanchors = n_anchors;
                        p.aligned_bases_q = n_abq;
                        p.aligned_bases_t = n_abt;
                        p.matched_bases = n_matched;
                        p.pident = pident;
                        p.qbegin = qb;
                        p.qend = qe;
                        p.tbegin = tbg;
                        p.tend = te;
                        res[nout] = p;
                    }
                    nout++;
                }
            }
        }
        if (i > lo) {
            if (lane == 0) {
                stack[sp] = lo;
                stack[sp + 1] = i;
            }
            sp += 2;
        }
        if (mi != hi - 1) {
            if (lane == 0) {
                stack[sp] = mi + 1;
                stack[sp + 1] = hi;
            }
            sp += 2;
        }
    }
    if (lane == 0) // stable sort by QBegin (lib-seq_compare.go:501-508)
        for (int i = 1; i < nout; i++) {
            LmChain2 x = res[i];
            int j = i - 1;
            while (j >= 0 && res[j].qbegin > x.qbegin) {
                res[j + 1] = res[j];
                j--;
            }
            res[j + 1] = x;
        }
    return nout;
}
