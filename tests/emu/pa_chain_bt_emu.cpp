// pa_chain_bt_emu.cpp - the wavefront backtrack of experiments/pa_chain_bt/pa_chain_bt.h on the host SIMT emulator against
// lm_run_chain2 (lm_algos.h): the chains it emits (every field, the double pident bit for bit), their number and order after
// the stable sort by QBegin.  Test infrastructure (tests/test_pa_chain_bt_emulated_cpu.py).
#include <stdint.h>
#include <string.h>

#include <vector>

#include "simt_emu.h"
#include "../../lexicmap_amd/csrc/lm_algos.h"

#define PCB_DEV static inline
#define PCB_LANE (simt::lane())
#define PCB_UNIFORM(x) ((int)simt::shfl((uint32_t)(x), 0, __LINE__)) /* lane 0's value in every lane */
#define PCB_LDS_SYNC() simt::wave_sync(__LINE__)
static inline unsigned long long emu_wave_max_u64(unsigned long long v, int site) {
    int p;
    const uint64_t *b = simt::rendezvous(false, v, site, &p);
    unsigned long long m = 0;
    for (int i = 0; i < 64; i++) m = b[i] > m ? b[i] : m;
    simt::leave(false, p);
    return m;
}
#define PCB_WAVE_MAX_U64(v) emu_wave_max_u64((v), __LINE__)
static long g_pcb_count[2]; // walk steps, tile loads (lane 0's)
#define PCB_COUNT(what) do { if (simt::lane() == 0) g_pcb_count[what]++; } while (0)

#include "../../lexicmap_amd/csrc/lm_pa_chain_bt_core.h"

static void sort_by_qbegin(LmChain2 *res, int nout) { // k_pa_chain_wave's stable sort (lib-seq_compare.go:501-508)
    for (int i = 1; i < nout; i++) {
        LmChain2 x = res[i];
        int j = i - 1;
        while (j >= 0 && res[j].qbegin > x.qbegin) {
            res[j + 1] = res[j];
            j--;
        }
        res[j + 1] = x;
    }
}

// returns the number of differences (0 = equal); *nchains: chains of the reference
extern "C" int pcb_emu_check(const int32_t *qb, const int32_t *tb, const uint8_t *len, int n, int max_gap, int band_base, int band_count, int min_score,
                             int min_align_len, double heuristic_pident, int *nchains) {
    std::vector<LmSub> a((size_t)n);
    for (int i = 0; i < n; i++) {
        memset(&a[i], 0, sizeof(LmSub));
        a[i].qbegin = qb[i];
        a[i].tbegin = tb[i];
        a[i].len = len[i];
    }
    LmChain2Opt opt;
    opt.max_gap = max_gap;
    opt.min_score = min_score;
    opt.min_align_len = min_align_len;
    opt.band_count = band_count;
    opt.band_base = band_base;
    opt.heuristic_pident = heuristic_pident;
    std::vector<uint64_t> msi((size_t)n);
    std::vector<int32_t> stack((size_t)2 * (n + 4)), stack2((size_t)2 * (n + 4), 0x5a5a5a5a);
    std::vector<LmChain2> ref((size_t)n + 1), got((size_t)n + 1);
    memset(ref.data(), 0, sizeof(LmChain2) * ref.size());
    memset(got.data(), 0, sizeof(LmChain2) * got.size());
    const int nref = lm_run_chain2(a.data(), n, opt, msi.data(), stack.data(), ref.data());
    sort_by_qbegin(ref.data(), nref);
    *nchains = nref;
    long long M = 0;
    int Mi = 0;
    for (int i = 1; i < n; i++)
        if ((long long)(msi[i] >> 32) > M) {
            M = (long long)(msi[i] >> 32);
            Mi = i;
        }
    PcbLds lds;
    memset(&lds, 0x5a, sizeof lds);
    int nout[64];
    g_pcb_count[0] = g_pcb_count[1] = 0;
    simt::run_wave([&](int lane) { nout[lane] = pa_chain_backtrack_wave(a.data(), n, opt, msi.data(), M, Mi, stack2.data(), got.data(), &lds); });
    int bad = 0;
    for (int l = 1; l < 64; l++) bad += nout[l] != nout[0];
    bad += nout[0] != nref;
    for (int i = 0; i < nref && i < nout[0]; i++) {
        const LmChain2 &x = ref[i], &y = got[i];
        bad += x.nanchors != y.nanchors || x.aligned_bases_q != y.aligned_bases_q || x.aligned_bases_t != y.aligned_bases_t ||
               x.matched_bases != y.matched_bases || memcmp(&x.pident, &y.pident, sizeof(double)) != 0 || x.qbegin != y.qbegin || x.qend != y.qend ||
               x.tbegin != y.tbegin || x.tend != y.tend;
    }
    return bad;
}
extern "C" void pcb_emu_counts(long *steps, long *tiles) {
    *steps = g_pcb_count[0];
    *tiles = g_pcb_count[1];
}

// ---- ClearSubstrPairs marks from LDS tiles (experiments/pa_chain_bt/pa_clear_tile.h) against lm_clear_sorted ----
#define PCC_DEV static inline
#define PCC_LANE (simt::lane())
#define PCC_LDS_SYNC() simt::wave_sync(__LINE__)
#include "../../lexicmap_amd/csrc/lm_pa_clear_tile.h"

// returns the number of marks that differ from lm_clear_sorted's; *kept: anchors the reference keeps
extern "C" int pcc_emu_check(const int32_t *qb, const int32_t *tb, const uint8_t *len, int n, int K, int *kept) {
    std::vector<LmSub> a((size_t)n), b;
    for (int i = 0; i < n; i++) {
        memset(&a[i], 0, sizeof(LmSub));
        a[i].qbegin = qb[i];
        a[i].tbegin = tb[i];
        a[i].len = len[i];
    }
    b = a;
    std::vector<uint8_t> mref((size_t)n + 1, 0), m((size_t)n + 1, 0x5a);
    *kept = lm_clear_sorted(b.data(), n, K, mref.data());
    PccLds lds;
    memset(&lds, 0x5a, sizeof lds);
    simt::run_wave([&](int) { pa_clear_marks_wave(a.data(), n, K, m.data(), &lds); });
    int bad = 0;
    for (int i = 0; i < n; i++) bad += (n > 1 ? mref[i] : 0) != m[i];
    return bad;
}
