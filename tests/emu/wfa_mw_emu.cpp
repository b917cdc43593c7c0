// wfa_mw_emu.cpp - the forward pass of wfa_mw_fwd.h (one alignment per workgroup of four wavefronts) on the host SIMT
// emulator, followed by the serial walk + replay of the backtrace rows it wrote: its alignments against the oracle's without a
// GPU.  Test infrastructure; built by tests/test_wfa_mw_emulated_cpu.py.
#include <stdint.h>
#include <string.h>

#include <vector>

#include "simt_emu.h"

#define WR_DEV static inline
#define WR_TID (simt::tid())
#define WR_BALLOT(p) simt::ballot((p), __LINE__)
#define WR_BARRIER() simt::barrier(__LINE__)
#define WR_UNIFORM(x) (x)
#define WR_CLZ(x) __builtin_clz(x)
#define WR_CLZLL(x) __builtin_clzll(x)
static inline uint32_t wr_pk_min_u16(uint32_t a, uint32_t b) {
    const uint32_t lo = (a & 0xffffu) < (b & 0xffffu) ? (a & 0xffffu) : (b & 0xffffu);
    const uint32_t hi = (a >> 16) < (b >> 16) ? (a >> 16) : (b >> 16);
    return lo | (hi << 16);
}
#define WR_WAVE_MIN_I32(v) \
    simt::wave_reduce((uint32_t)(v), __LINE__, [](uint32_t a, uint32_t b) { return (uint32_t)((int32_t)a < (int32_t)b ? (int32_t)a : (int32_t)b); })
#define WR_WAVE_PKMIN_U16(v) simt::wave_reduce((v), __LINE__, [](uint32_t a, uint32_t b) { return wr_pk_min_u16(a, b); })

#include "../../lexicmap_amd/csrc/lm_wfa_mw_fwd.h"
#include "wfa_host_walk.h"

template <int NCW, bool WIN>
static long run1(const uint8_t *q, int qlen, const uint8_t *t, int tlen, int seq_words, int max_score, int arena_cap, uint64_t *ops, int ops_cap,
                 WrEmuOut *out) {
    constexpr int W = MW_THREADS * NCW;
    std::vector<int32_t> hdr((size_t)max_score + 8, 0), ring((size_t)9 * W, 0x5a5a5a5a), red(40, 0x5a5a5a5a);
    std::vector<uint8_t> bt((size_t)arena_cap + 16, 0xff);
    const size_t sw = WIN ? MW_WINW + 2 : (size_t)seq_words + 2;
    std::vector<uint32_t> qb(sw, 0xdeadbeefu), tb(sw, 0xdeadbeefu);
    MwProb p;
    p.q = q;
    p.t = t;
    p.plen = qlen;
    p.tlen = tlen;
    p.hdr2 = hdr.data();
    p.bt = bt.data();
    p.arena_cap = arena_cap;
    p.max_score = max_score;
    MwLds L;
    L.ring = ring.data();
    L.qbuf = qb.data();
    L.tbuf = tb.data();
    L.red = red.data();
    std::vector<MwRes> res(MW_THREADS);
    const long ncoll = simt::run_block(MW_WAVES, [&](int tid) { wfa_mw_forward<NCW, WIN>(p, L, seq_words, &res[tid]); });
    for (int i = 1; i < MW_THREADS; i++)
        if (memcmp(&res[0], &res[i], sizeof(MwRes)) != 0) {
            fprintf(stderr, "wfa_mw_emu: threads disagree on the result\n");
            abort();
        }
    memset(out, 0, sizeof *out);
    out->status = res[0].status;
    out->score = res[0].score;
    out->used = res[0].used;
    if (out->status == 0 && walk_replay(hdr.data(), bt.data(), out->score, q, qlen, t, tlen, ops, ops_cap, out) != 0) out->status = 1;
    return ncoll;
}

extern "C" long mw_emu_run(int ncw, int win, const uint8_t *q, int qlen, const uint8_t *t, int tlen, int seq_words, int max_score, int arena_cap,
                           uint64_t *ops, int ops_cap, WrEmuOut *out) {
    switch (ncw * 2 + (win ? 1 : 0)) {
    case 2: return run1<1, false>(q, qlen, t, tlen, seq_words, max_score, arena_cap, ops, ops_cap, out);
    case 3: return run1<1, true>(q, qlen, t, tlen, seq_words, max_score, arena_cap, ops, ops_cap, out);
    case 4: return run1<2, false>(q, qlen, t, tlen, seq_words, max_score, arena_cap, ops, ops_cap, out);
    case 5: return run1<2, true>(q, qlen, t, tlen, seq_words, max_score, arena_cap, ops, ops_cap, out);
    case 8: return run1<4, false>(q, qlen, t, tlen, seq_words, max_score, arena_cap, ops, ops_cap, out);
    case 9: return run1<4, true>(q, qlen, t, tlen, seq_words, max_score, arena_cap, ops, ops_cap, out);
    }
    return -1;
}
