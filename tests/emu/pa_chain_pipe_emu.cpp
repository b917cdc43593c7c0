// pa_chain_pipe_emu.cpp - experiments/pa_chain_pipe/pa_chain_pipe.h (STAGED: a workgroup of wavefronts pipelined over the
// anchors of one chaining window) on the host SIMT emulator against lm_run_chain2 (lm_algos.h).  Test infrastructure.
#include <stdint.h>
#include <string.h>

#include <vector>

#include "simt_emu.h"
#include "../../lexicmap_amd/csrc/lm_algos.h"

#define PCP_DEV static inline
#define PCP_TID (simt::tid())
#define PCP_BALLOT(p) simt::ballot((p), __LINE__)
#define PCP_WAVE_SYNC() simt::wave_sync(__LINE__)
#define PCP_BARRIER() simt::barrier(__LINE__)
#define PCP_POPCLL(x) __builtin_popcountll(x)
#define PCP_FFSLL(x) __builtin_ffsll((long long)(x))
static inline unsigned long long emu_wave_max_u64(unsigned long long v, int site) {
    int p;
    const uint64_t *b = simt::rendezvous(false, v, site, &p);
    unsigned long long m = 0;
    for (int i = 0; i < 64; i++) m = b[i] > m ? b[i] : m;
    simt::leave(false, p);
    return m;
}
#define PCP_WAVE_MAX_U64(v) emu_wave_max_u64((v), __LINE__)
#define PCP_BCAST32(v, l) simt::shfl((uint32_t)(v), (l), __LINE__)
// a wave-uniform read: every lane of the wavefront gets what lane 0 read (on the device: a scalar load / readfirstlane)
static inline int emu_load_done(volatile int *p, int site) { return (int)simt::shfl((uint32_t)*p, 0, site); }
#define PCP_LOAD_DONE(p) emu_load_done((p), __LINE__)
#define PCP_STORE_DONE(p, v) (*(p) = (v))
static long g_spins = 0;
static inline void emu_spin() { // let the other wavefronts run; a pipeline that cannot make progress must not hang the test
    if (++g_spins > 400000000L) simt::fail("the pipeline does not make progress", 0, 0);
    simt::yield_next();
}
#define PCP_SPIN() emu_spin()
#define PCP_GLOBAL_FENCE() simt::wave_sync(__LINE__)
#define PCP_LOAD_MSI(p) (*(p))
// scheduling perturbation: at these points a lane may hand the processor to the other lanes a random number of times, so that
// the wavefronts of the pipeline interleave differently from run to run (the emulator's own order is a fixed round robin)
static uint64_t g_sched = 0;
static inline void emu_sched_point() {
    if (!g_sched) return;
    g_sched = g_sched * 6364136223846793005ull + 1442695040888963407ull;
    for (int k = (int)((g_sched >> 60) & 7); k > 0; k--) simt::yield_next();
}
#define PCP_SCHED_POINT() emu_sched_point()

#include "../../lexicmap_amd/csrc/lm_pa_chain_pipe_dp.h"

// returns 0 when the emulated pipeline's msi / best score / best anchor equal lm_run_chain2's
extern "C" int pcp_emu_check(const int32_t *qb, const int32_t *tb, const uint8_t *len, int n, int max_gap, int band_base, int band_count,
                             long long *M_out, int *Mi_out, long *collectives, unsigned long long sched_seed) {
    g_sched = sched_seed;
    std::vector<LmSub> a((size_t)n);
    for (int i = 0; i < n; i++) {
        memset(&a[i], 0, sizeof(LmSub));
        a[i].qbegin = qb[i];
        a[i].tbegin = tb[i];
        a[i].len = len[i];
    }
    LmChain2Opt opt;
    opt.max_gap = max_gap;
    opt.min_score = 1 << 30; // the reference run stops after the DP: only msi is compared
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
    PcpLds lds;
    memset(&lds, 0x5a, sizeof lds);
    std::vector<long long> M((size_t)64 * PCP_NW);
    std::vector<int> Mi((size_t)64 * PCP_NW);
    g_spins = 0;
    const long nc = simt::run_block(PCP_NW, [&](int t) { pa_chain_dp_pipe(a.data(), n, opt, msi.data(), &lds, &M[t], &Mi[t]); }, 1 << 18);
    int bad = 0;
    for (int l = 1; l < 64 * PCP_NW; l++) bad += M[l] != M[0] || Mi[l] != Mi[0];
    for (int i = 0; i < n; i++) bad += msi[i] != msi_ref[i];
    if (n >= 2) bad += (M[0] != Mref) || (Mi[0] != Miref);
    if (M_out) *M_out = M[0];
    if (Mi_out) *Mi_out = Mi[0];
    if (collectives) *collectives = nc;
    return bad;
}
