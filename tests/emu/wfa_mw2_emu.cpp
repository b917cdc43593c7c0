// wfa_mw2_emu.cpp - the workgroup forward pass of experiments/wfa_lean2/wfa_mw2_fwd.h (four wavefronts per alignment, the
// restructuring of lm_wfa_mw_fwd.h) on the host SIMT emulator, followed by the serial walk + replay of the backtrace rows it
// wrote: its alignments against the oracle's without a GPU.  Test infrastructure; built by tests/test_wfa_mw2_emulated_cpu.py.
#include <stdint.h>
#include <string.h>

#include <vector>

#include "simt_emu.h"

#define WR_DEV static inline
#define WR_LDS
#define WR_TID (simt::tid())
#define WR_BALLOT(p) simt::ballot((p), __LINE__)
#define WR_WAVE_SYNC() simt::wave_sync(__LINE__)
#define WR_BARRIER() simt::barrier(__LINE__)
#define WR_UNIFORM(x) (x)
#define WR_CLZ(x) ((x) ? __builtin_clz(x) : 32)
#define WR_ALIGNBIT(hi, lo, sh) ((uint32_t)(((((uint64_t)(hi)) << 32) | (uint64_t)(lo)) >> ((sh) & 31)))
#define WR_FF1(x) ((x) ? __builtin_ctzll(x) : -1) /* s_ff1_i32_b64 */
#define WR_FLB(x) ((x) ? __builtin_clzll(x) : -1) /* s_flbit_i32_b64 */
#define WR_READLANE(v, l) ((int32_t)simt::shfl((uint32_t)(v), (l), __LINE__))
#define WR_WAVE_MIN_I32(v) \
    ((int32_t)simt::wave_reduce((uint32_t)(v), __LINE__, [](uint32_t a, uint32_t b) { return (uint32_t)((int32_t)a < (int32_t)b ? (int32_t)a : (int32_t)b); }))
static inline int32_t emu_quad_min(int32_t v, int site) { // the minimum over the four lanes of a quad, in each of them
    int32_t x = (int32_t)simt::shfl((uint32_t)v, simt::lane() ^ 1, site);
    v = x < v ? x : v;
    x = (int32_t)simt::shfl((uint32_t)v, simt::lane() ^ 2, site + 100000);
    return x < v ? x : v;
}
#define WR_QUAD_MIN_I32(v) emu_quad_min((v), __LINE__)

#include "../../lexicmap_amd/csrc/lm_wfa_mw2_fwd.h"
#include "wfa_host_walk.h"

static bool pack_seq(const uint8_t *s, int n, std::vector<uint32_t> &w) {
    w.assign((size_t)(n + 15) / 16 + 3, 0u); // word 0 of the sequence is w[1]: one readable word in front
    w[0] = 0xdeadbeefu;
    bool ok = true;
    for (int i = 0; i < n; i++) {
        const uint32_t c = s[i], code = (c >> 1) & 3u;
        ok = ok && c == ((0x47544341u >> (code << 3)) & 0xffu);
        w[1 + (i >> 4)] |= code << (30 - 2 * (i & 15));
    }
    return ok;
}

template <int NCW, bool WIN>
static long run1(const uint8_t *q, int qlen, const uint8_t *t, int tlen, int max_score, int arena_cap, uint64_t *ops, int ops_cap, WrEmuOut *out,
                 int *recentres) {
    std::vector<int32_t> hdr((size_t)max_score + 8, 0), ring((size_t)mw2_ring_cells<NCW>(), 0x5a5a5a5a), red(MW2_RED_WORDS, 0x5a5a5a5a);
    std::vector<uint8_t> bt((size_t)arena_cap + 16, 0xff);
    std::vector<uint32_t> qb, tb;
    memset(out, 0, sizeof *out);
    *recentres = 0;
    if (WIN) {
        qb.assign((size_t)L2_WINW + 3, 0xdeadbeefu);
        tb.assign((size_t)L2_WINW + 3, 0xdeadbeefu);
    } else if (!pack_seq(q, qlen, qb) || !pack_seq(t, tlen, tb)) {
        out->status = 3;
        return 1;
    }
    L2Prob p;
    p.q = q;
    p.t = t;
    p.plen = qlen;
    p.tlen = tlen;
    p.hdr2 = hdr.data();
    p.bt = bt.data();
    p.arena_cap = arena_cap;
    p.max_score = max_score;
    std::vector<L2Res> res(MW2_THREADS);
    const long ncoll = simt::run_block(4, [&](int tid) { wfa_mw2_forward<NCW, WIN>(p, ring.data(), qb.data() + 1, tb.data() + 1, red.data(), &res[tid]); });
    for (int i = 1; i < MW2_THREADS; i++)
        if (memcmp(&res[0], &res[i], sizeof(L2Res)) != 0) {
            fprintf(stderr, "wfa_mw2_emu: threads disagree on the result\n");
            abort();
        }
    out->status = res[0].status;
    out->score = res[0].score;
    out->used = res[0].used;
    *recentres = res[0].recentres;
    if (out->status == 0 && walk_replay(hdr.data(), bt.data(), out->score, q, qlen, t, tlen, ops, ops_cap, out) != 0) out->status = 1;
    return ncoll;
}

extern "C" long mw2_emu_run(int ncw, int win, const uint8_t *q, int qlen, const uint8_t *t, int tlen, int max_score, int arena_cap, uint64_t *ops,
                            int ops_cap, WrEmuOut *out, int *recentres) {
    switch (ncw * 2 + (win ? 1 : 0)) {
    case 2: return run1<1, false>(q, qlen, t, tlen, max_score, arena_cap, ops, ops_cap, out, recentres);
    case 3: return run1<1, true>(q, qlen, t, tlen, max_score, arena_cap, ops, ops_cap, out, recentres);
    case 4: return run1<2, false>(q, qlen, t, tlen, max_score, arena_cap, ops, ops_cap, out, recentres);
    case 5: return run1<2, true>(q, qlen, t, tlen, max_score, arena_cap, ops, ops_cap, out, recentres);
    case 8: return run1<4, false>(q, qlen, t, tlen, max_score, arena_cap, ops, ops_cap, out, recentres);
    case 9: return run1<4, true>(q, qlen, t, tlen, max_score, arena_cap, ops, ops_cap, out, recentres);
    }
    return -1;
}
