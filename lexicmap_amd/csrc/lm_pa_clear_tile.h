// lm_pa_clear_tile.h - the marks of ClearSubstrPairs (lib-index-search.go:927-972 as lm_clear_sorted states
// it: an anchor is dropped when it lies inside an EARLIER anchor of the sorted list whose QBegin is at most K - len before its
// own) from LDS tiles.  Equal to lm_clear_sorted on the host SIMT emulator; product code since round 5.
//
// k_pa_chain_wave gives every anchor a lane that binary-searches the list in GLOBAL memory for its first candidate (~11
// dependent loads) and then reads the candidates one by one (a 16-byte load each): ~20 dependent round trips per 64 anchors,
// the other long phase of the kernel beside the backtrack.  The candidates of anchor i are the anchors right before it - those
// with QBegin >= QEnd(i) - K, at most ~20 bases back - so the wavefront loads the 64 anchors of a pass and the 64 before them
// into LDS once (one coalesced round trip) and every lane scans backwards from its own anchor until a QBegin falls below its
// bound; a lane whose candidates reach beyond the halo goes on in global memory (rare: more than 64 anchors within 20 bases).
#pragma once
#include <stdint.h>

#define PCC_TILE 128 /* the 64 anchors of a pass + a halo of 64 before them */
struct PccLds {
    int32_t q[PCC_TILE], qe[PCC_TILE], t[PCC_TILE], te[PCC_TILE];
};

// sb[0..n): the window's anchors, sorted; marks[i] = 1 when anchor i is nested in an earlier one.  All 64 lanes call.
PCC_DEV void pa_clear_marks_wave(const LmSub *sb, int n, int K, uint8_t *marks, PccLds *L) {
    const int lane = PCC_LANE;
    for (int c = 0; c < n; c += 64) {
        const int tb = c >= 64 ? c - 64 : 0; // first anchor of the tile
        PCC_LDS_SYNC();                      // every lane is done with the previous tile
        for (int g = tb + lane; g < c + 64 && g < n; g += 64) {
            const LmSub s = sb[g];
            L->q[g - tb] = s.qbegin;
            L->qe[g - tb] = s.qbegin + (int32_t)s.len;
            L->t[g - tb] = s.tbegin;
            L->te[g - tb] = s.tbegin + (int32_t)s.len;
        }
        PCC_LDS_SYNC();
        const int i = c + lane;
        if (i < n) {
            uint8_t mk = 0;
            if (i >= 1) {
                const int o = i - tb;
                const int32_t vqend = L->qe[o], vt = L->t[o], vte = L->te[o];
                int32_t upbound = vqend - K;
                if (upbound < 0) upbound = 0;
                int j = i - 1;
                for (; j >= tb; j--) { // (the list is sorted by QBegin: the candidates are a run right before i)
                    const int oj = j - tb;
                    if (L->q[oj] < upbound) break;
                    if (vqend <= L->qe[oj] && vt >= L->t[oj] && vte <= L->te[oj]) {
                        mk = 1;
                        break;
                    }
                }
                if (!mk && j < tb && tb > 0) { // the halo did not reach the first candidate
                    for (; j >= 0; j--) {
                        const LmSub p = sb[j];
                        if (p.qbegin < upbound) break;
                        if (vqend <= p.qbegin + (int32_t)p.len && vt >= p.tbegin && vte <= p.tbegin + (int32_t)p.len) {
                            mk = 1;
                            break;
                        }
                    }
                }
            }
            marks[i] = mk;
        }
    }
}
