// wfa_row_emu.cpp - the forward pass of wfa_row_fwd.h run on the host SIMT emulator (simt_emu.h), followed by a plain serial
// walk + replay of the backtrace rows it wrote (same byte format as k_wfa_lean's), so that its alignments can be compared
// with the oracle's on a machine without a GPU.  Test infrastructure; built by tests/test_wfa_row_emulated_cpu.py.
#include <stdint.h>
#include <string.h>

#include <vector>

#include "simt_emu.h"

#define WR_DEV static inline
#define WR_LANE (simt::lane())
#define WR_BALLOT(p) simt::ballot((p), __LINE__)
#define WR_SHFL(v, src) simt::shfl((v), (src), __LINE__)
#define WR_LDS_SYNC() simt::wave_sync(__LINE__)
#define WR_CLZ(x) __builtin_clz(x)
static inline uint32_t wr_pk_min_u16(uint32_t a, uint32_t b) {
    const uint32_t lo = (a & 0xffffu) < (b & 0xffffu) ? (a & 0xffffu) : (b & 0xffffu);
    const uint32_t hi = (a >> 16) < (b >> 16) ? (a >> 16) : (b >> 16);
    return lo | (hi << 16);
}
#define WR_ROW_MIN_I32(v) \
    simt::row_reduce((uint32_t)(v), __LINE__, [](uint32_t a, uint32_t b) { return (uint32_t)((int32_t)a < (int32_t)b ? (int32_t)a : (int32_t)b); })
#define WR_ROW_PKMIN_U16(v) simt::row_reduce((v), __LINE__, [](uint32_t a, uint32_t b) { return wr_pk_min_u16(a, b); })

#include "../../experiments/wfa_row/wfa_row_fwd.h"

#include "wfa_host_walk.h"

template <int NCR>
static long run4(const uint8_t *const *q, const int32_t *qlen, const uint8_t *const *t, const int32_t *tlen, int nvalid, int seq_words,
                 int max_score, int arena_cap, uint64_t *const *ops, int ops_cap, WrEmuOut *out) {
    std::vector<std::vector<int32_t>> hdr(4, std::vector<int32_t>((size_t)max_score + 8, 0));
    std::vector<std::vector<uint8_t>> bt(4, std::vector<uint8_t>((size_t)arena_cap + 16, 0xff));
    std::vector<std::vector<uint32_t>> lds(8, std::vector<uint32_t>((size_t)seq_words + 2, 0xdeadbeefu));
    static const uint8_t none[1] = {0};
    WrRow rows[4];
    for (int r = 0; r < 4; r++) {
        const bool v = r < nvalid;
        rows[r].q = v ? q[r] : none;
        rows[r].t = v ? t[r] : none;
        rows[r].plen = v ? qlen[r] : 0;
        rows[r].tlen = v ? tlen[r] : 0;
        rows[r].hdr2 = hdr[r].data();
        rows[r].bt = bt[r].data();
        rows[r].arena_cap = arena_cap;
        rows[r].max_score = max_score;
        rows[r].qbuf = lds[2 * r].data();
        rows[r].tbuf = lds[2 * r + 1].data();
        rows[r].valid = v ? 1 : 0;
    }
    WrRes res[64];
    const long ncoll = simt::run_wave([&](int lane) {
        WrRow p = rows[lane >> 4];
        wfa_row4_forward<NCR>(p, seq_words, &res[lane]);
    });
    for (int r = 0; r < nvalid; r++) {
        for (int l = 1; l < 16; l++) // row-uniform results
            if (memcmp(&res[16 * r], &res[16 * r + l], sizeof(WrRes)) != 0) {
                fprintf(stderr, "wfa_row_emu: lanes of row %d disagree\n", r);
                abort();
            }
        WrEmuOut &o = out[r];
        memset(&o, 0, sizeof o);
        o.status = res[16 * r].status;
        o.score = res[16 * r].score;
        o.used = res[16 * r].used;
        if (o.status == 0 && walk_replay(hdr[r].data(), bt[r].data(), o.score, q[r], qlen[r], t[r], tlen[r], ops[r], ops_cap, &o) != 0) o.status = 1;
    }
    return ncoll;
}

extern "C" long wr_emu_run4(int ncr, const uint8_t *const *q, const int32_t *qlen, const uint8_t *const *t, const int32_t *tlen, int nvalid,
                            int seq_words, int max_score, int arena_cap, uint64_t *const *ops, int ops_cap, WrEmuOut *out) {
    switch (ncr) {
    case 2: return run4<2>(q, qlen, t, tlen, nvalid, seq_words, max_score, arena_cap, ops, ops_cap, out);
    case 4: return run4<4>(q, qlen, t, tlen, nvalid, seq_words, max_score, arena_cap, ops, ops_cap, out);
    case 8: return run4<8>(q, qlen, t, tlen, nvalid, seq_words, max_score, arena_cap, ops, ops_cap, out);
    }
    return -1;
}
