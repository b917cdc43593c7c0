// wfa_lean2_emu.cpp - the single-wavefront forward pass of experiments/wfa_lean2/wfa_lean2_fwd.h (staged restructuring of
// k_wfa_lean: no-wrap ring, ballot trimming, fused extension) on the host SIMT emulator, followed by the serial walk + replay of
// the backtrace rows it wrote: its alignments against the oracle's without a GPU.  Test infrastructure; built by
// tests/test_wfa_lean2_emulated_cpu.py.
#include <stdint.h>
#include <string.h>

#include <vector>

#include "simt_emu.h"

#define WR_DEV static inline
#define WR_TID (simt::tid())
#define WR_BALLOT(p) simt::ballot((p), __LINE__)
#define WR_WAVE_SYNC() simt::wave_sync(__LINE__)
#define WR_UNIFORM(x) (x)
#define WR_LDS
#define WR_CLZ(x) ((x) ? __builtin_clz(x) : 32)
#define WR_ALIGNBIT(hi, lo, sh) ((uint32_t)(((((uint64_t)(hi)) << 32) | (uint64_t)(lo)) >> ((sh) & 31)))
#define WR_FF1(x) ((x) ? __builtin_ctzll(x) : -1) /* s_ff1_i32_b64 */
#define WR_FLB(x) ((x) ? __builtin_clzll(x) : -1) /* s_flbit_i32_b64 */
#define WR_READLANE(v, l) ((int32_t)simt::shfl((uint32_t)(v), (l), __LINE__))
#define WR_WAVE_MIN_I32(v) \
    ((int32_t)simt::wave_reduce((uint32_t)(v), __LINE__, [](uint32_t a, uint32_t b) { return (uint32_t)((int32_t)a < (int32_t)b ? (int32_t)a : (int32_t)b); }))

// dynamic counts of the forward pass (lane 0 counts; l2_emu_counts reads and clears): steps per (edge, flavour), extension
// passes, cut-offs, the sum of the row widths
struct L2Counts {
    long step[2][5];
    long ext_pass, cutoff, width;
};
static L2Counts g_cnt;
#define L2_COUNT(what, n) \
    do {                  \
        if (simt::tid() == 0) g_cnt.what += (n); \
    } while (0)
#include "../../lexicmap_amd/csrc/lm_wfa_lean2_fwd.h"
#include "wfa_host_walk.h"

// 2-bit packing of k_wfa_lean's pack16 ('A' 0, 'C' 1, 'T' 2, 'G' 3; first base in the top bits); false: not plain ACGT
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

template <int NC, typename RT, bool WIN = false, int MARGIN = L2_SHRINK_MARGIN>
static long run1(const uint8_t *q, int qlen, const uint8_t *t, int tlen, int max_score, int arena_cap, uint64_t *ops, int ops_cap, WrEmuOut *out,
                 int *recentres) {
    std::vector<int32_t> hdr((size_t)max_score + 8, 0);
    std::vector<uint8_t> bt((size_t)arena_cap + 16, 0xff);
    std::vector<RT> ring((size_t)l2_ring_cells<NC>(), (RT)0x5a5a);
    std::vector<uint32_t> qb, tb;
    memset(out, 0, sizeof *out);
    *recentres = 0;
    if (WIN) { // the windows: filled by the forward pass itself
        qb.assign((size_t)L2_WINW + 3, 0xdeadbeefu);
        tb.assign((size_t)L2_WINW + 3, 0xdeadbeefu);
    } else if (!pack_seq(q, qlen, qb) || !pack_seq(t, tlen, tb)) { // (the kernel's packer says so: status 3)
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
    std::vector<L2Res> res(64);
    const long ncoll = simt::run_wave([&](int lane) { wfa_lean2_forward<NC, RT, WIN, MARGIN>(p, ring.data(), qb.data() + 1, tb.data() + 1, &res[lane]); });
    for (int i = 1; i < 64; i++)
        if (memcmp(&res[0], &res[i], sizeof(L2Res)) != 0) {
            fprintf(stderr, "wfa_lean2_emu: lanes disagree on the result\n");
            abort();
        }
    out->status = res[0].status;
    out->score = res[0].score;
    out->used = res[0].used;
    *recentres = res[0].recentres;
    if (out->status == 0 && walk_replay(hdr.data(), bt.data(), out->score, q, qlen, t, tlen, ops, ops_cap, out) != 0) out->status = 1;
    return ncoll;
}

extern "C" void l2_emu_counts(long *out) { // 13 longs
    memcpy(out, &g_cnt, sizeof g_cnt);
    memset(&g_cnt, 0, sizeof g_cnt);
}
extern "C" long l2_emu_run(int nc, int r16, int win, const uint8_t *q, int qlen, const uint8_t *t, int tlen, int max_score, int arena_cap,
                           uint64_t *ops, int ops_cap, WrEmuOut *out, int *recentres) {
    if (win) switch (nc) {
        case 1: return run1<1, int32_t, true>(q, qlen, t, tlen, max_score, arena_cap, ops, ops_cap, out, recentres);
        case 2: return run1<2, int32_t, true>(q, qlen, t, tlen, max_score, arena_cap, ops, ops_cap, out, recentres);
        case 4: return run1<4, int32_t, true>(q, qlen, t, tlen, max_score, arena_cap, ops, ops_cap, out, recentres);
        case 8: return run1<8, int32_t, true>(q, qlen, t, tlen, max_score, arena_cap, ops, ops_cap, out, recentres);
        case 16: return run1<16, int32_t, true>(q, qlen, t, tlen, max_score, arena_cap, ops, ops_cap, out, recentres);
        default: return -1;
        }
    switch (nc * 2 + (r16 ? 1 : 0)) {
    case 2: return run1<1, int32_t>(q, qlen, t, tlen, max_score, arena_cap, ops, ops_cap, out, recentres);
    case 3: return run1<1, int16_t>(q, qlen, t, tlen, max_score, arena_cap, ops, ops_cap, out, recentres);
    case 4: return run1<2, int32_t>(q, qlen, t, tlen, max_score, arena_cap, ops, ops_cap, out, recentres);
    case 5: return run1<2, int16_t>(q, qlen, t, tlen, max_score, arena_cap, ops, ops_cap, out, recentres);
    case 8: return run1<4, int32_t>(q, qlen, t, tlen, max_score, arena_cap, ops, ops_cap, out, recentres);
    case 9: return run1<4, int16_t>(q, qlen, t, tlen, max_score, arena_cap, ops, ops_cap, out, recentres);
    case 16: return run1<8, int32_t>(q, qlen, t, tlen, max_score, arena_cap, ops, ops_cap, out, recentres);
    case 32: return run1<16, int32_t>(q, qlen, t, tlen, max_score, arena_cap, ops, ops_cap, out, recentres);
    }
    return -1;
}
