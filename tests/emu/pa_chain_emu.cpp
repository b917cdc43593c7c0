// pa_chain_emu.cpp - pa_chain_dp.h on the host SIMT emulator against lm_run_chain2 (lm_algos.h, the CPU-checked statement of
// the device logic): scores, predecessors, best score and its anchor.  Test infrastructure (tests/test_pa_chain_emulated_cpu.py).
#include <stdint.h>
#include <string.h>

#include <vector>

#include "simt_emu.h"
#include "../../lexicmap_amd/csrc/lm_algos.h"

#define PCD_DEV static inline
#define PCD_LANE (simt::lane())
#define PCD_BALLOT(p) simt::ballot((p), __LINE__)
#define PCD_LDS_SYNC() simt::wave_sync(__LINE__)
#define PCD_GLOBAL_FENCE() simt::wave_sync(__LINE__)
#define PCD_POPCLL(x) __builtin_popcountll(x)
#define PCD_FFSLL(x) __builtin_ffsll((long long)(x))
static inline unsigned long long emu_wave_max_u64(unsigned long long v, int site) {
    // two 32-bit rendezvous (the emulator's exchange word holds 64 bits: one is enough)
    int p;
    const uint64_t *b = simt::rendezvous(false, v, site, &p);
    unsigned long long m = 0;
    for (int i = 0; i < 64; i++) m = b[i] > m ? b[i] : m;
    simt::leave(false, p);
    return m;
}
#define PCD_WAVE_MAX_U64(v) emu_wave_max_u64((v), __LINE__)
static inline int emu_shift_in(int newv, int v, int site) { // lane 0 <- newv, lane l <- lane l - 1
    const int l = simt::lane();
    const uint32_t got = simt::shfl((uint32_t)v, l > 0 ? l - 1 : 0, site);
    return l > 0 ? (int)got : newv;
}
#define PCD_SHIFT_IN(newv, v) emu_shift_in((int)(newv), (int)(v), __LINE__)
#define PCD_WAVE_MAX_I32(v) ((int)simt::wave_reduce((uint32_t)(v), __LINE__, [](uint32_t a, uint32_t b) { return (int)a > (int)b ? a : b; }))
#define PCD_CLZLL(x) __builtin_clzll((unsigned long long)(x))

#include "../../lexicmap_amd/csrc/lm_pa_chain_dp_core.h"

// subs: n anchors {qbegin, tbegin, len} (sorted the way the kernel gets them); returns 0 when the emulated DP equals lm_run_chain2's
extern "C" int pcd_emu_check(const int32_t *qb, const int32_t *tb, const uint8_t *len, int n, int max_gap, int band_base, int band_count,
                             uint64_t *msi_out, long long *M_out, int *Mi_out) {
    std::vector<LmSub> a((size_t)n);
    for (int i = 0; i < n; i++) {
        memset(&a[i], 0, sizeof(LmSub));
        a[i].qbegin = qb[i];
        a[i].tbegin = tb[i];
        a[i].len = len[i];
    }
    LmChain2Opt opt;
    opt.max_gap = max_gap;
    opt.min_score = 1 << 30; // the reference run stops after the DP (nothing reaches this score): only msi is compared
    opt.min_align_len = 0;
    opt.band_count = band_count;
    opt.band_base = band_base;
    opt.heuristic_pident = 0;
    std::vector<uint64_t> msi_ref((size_t)n), msi((size_t)n, 0xdeadbeefdeadbeefull);
    std::vector<int32_t> stack((size_t)2 * (n + 2));
    std::vector<LmChain2> out((size_t)n + 1);
    lm_run_chain2(a.data(), n, opt, msi_ref.data(), stack.data(), out.data());
    long long Mref = 0;
    int Miref = 0;
    for (int i = 1; i < n; i++)
        if ((long long)(msi_ref[i] >> 32) > Mref) {
            Mref = (long long)(msi_ref[i] >> 32);
            Miref = i;
        }
    PcdLds lds;
    memset(&lds, 0x5a, sizeof lds);
    long long M[64];
    int Mi[64];
    simt::run_wave([&](int lane) { pa_chain_dp_reg(a.data(), n, opt, msi.data(), &lds, &M[lane], &Mi[lane]); });
    int bad = 0;
    for (int l = 1; l < 64; l++) bad += M[l] != M[0] || Mi[l] != Mi[0];
    // (lm_run_chain2 seeds msi[0] with predecessor 0 and M with 0: the kernel's loop starts at anchor 1 the same way)
    for (int i = 0; i < n; i++) bad += msi[i] != msi_ref[i];
    if (n >= 2) bad += (M[0] != Mref) || (Mi[0] != Miref);
    if (msi_out) memcpy(msi_out, msi.data(), sizeof(uint64_t) * n);
    if (M_out) *M_out = M[0];
    if (Mi_out) *Mi_out = Mi[0];
    return bad;
}
