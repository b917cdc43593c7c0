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
#define WR_LDS_SYNC() simt::barrier(__LINE__)
#define WR_CLZ(x) __builtin_clz(x)
static inline uint32_t wr_pk_min_u16(uint32_t a, uint32_t b) {
    const uint32_t lo = (a & 0xffffu) < (b & 0xffffu) ? (a & 0xffffu) : (b & 0xffffu);
    const uint32_t hi = (a >> 16) < (b >> 16) ? (a >> 16) : (b >> 16);
    return lo | (hi << 16);
}
#define WR_ROW_MIN_I32(v) \
    simt::row_reduce((uint32_t)(v), __LINE__, [](uint32_t a, uint32_t b) { return (uint32_t)((int32_t)a < (int32_t)b ? (int32_t)a : (int32_t)b); })
#define WR_ROW_PKMIN_U16(v) simt::row_reduce((v), __LINE__, [](uint32_t a, uint32_t b) { return wr_pk_min_u16(a, b); })

#include "wfa_row_fwd.h"

struct WrEmuOut {
    int32_t status, score, nops, qbegin, qend, tbegin, tend;
    uint32_t align_len, matches, gaps, gap_regions;
    int32_t used;
};

// serial statement of bt_walk + bt_replay (lm_kernels.hip): edit operations from the backtrace bytes, match runs by greedy
// extension, run list and statistics of the M-trimmed alignment
static int walk_replay(const int32_t *hdr2, const uint8_t *bt, int s_final, const uint8_t *q, int plen, const uint8_t *t, int tlen,
                       uint64_t *ops, int ops_cap, WrEmuOut *o) {
    std::vector<uint8_t> rev;
    int score = s_final, k = tlen - plen, matrix = 0;
    while (score > 0) {
        const int e = score >> 1;
        const int rlo = hdr2[2 * e], rbase = hdr2[2 * e + 1], rend = hdr2[2 * (e + 1) + 1];
        if (k < rlo || k - rlo >= rend - rbase) return -1;
        const int code = bt[rbase + (k - rlo)];
        int op, ext;
        if (matrix == 0) {
            op = code & 3;
            ext = op == 1 ? (code >> 2) & 1 : (code >> 3) & 1;
        } else {
            op = matrix;
            ext = matrix == 1 ? (code >> 2) & 1 : (code >> 3) & 1;
        }
        if (op == 3) return -1;
        rev.push_back((uint8_t)(op | (matrix == 0 ? 4 : 0)));
        if (op == 0) {
            score -= 4;
            matrix = 0;
        } else {
            score -= ext ? 2 : 8;
            k += op == 1 ? -1 : 1;
            matrix = ext ? op : 0;
        }
    }
    if (score != 0 || k != 0 || matrix != 0) return -1;
    int v = 0, h = 0, cur_op = 0, cur_n = 0, run_q = 0, run_t = 0, wp = 0;
    bool seen_m = false, overflow = false;
    int alen = 0, matches = 0, gaps = 0, greg = 0, c_alen = 0, c_matches = 0, c_gaps = 0, c_greg = 0;
    int qbegin = 0, tbegin = 0, qend = 0, tend = 0;
    auto flush = [&]() {
        if (cur_n == 0) return;
        if (wp >= ops_cap)
            overflow = true;
        else
            ops[wp] = ((uint64_t)(uint32_t)cur_op << 32) | (uint32_t)cur_n;
        wp++;
        if (cur_op == 'M') {
            if (!seen_m) {
                seen_m = true;
                qbegin = run_q + 1;
                tbegin = run_t + 1;
            }
            alen += cur_n;
            matches += cur_n;
            c_alen = alen;
            c_matches = matches;
            c_gaps = gaps;
            c_greg = greg;
            qend = run_q + cur_n;
            tend = run_t + cur_n;
        } else if (seen_m) {
            alen += cur_n;
            if (cur_op != 'X') {
                gaps += cur_n;
                greg++;
            }
        }
    };
    auto extend = [&]() {
        int run = 0;
        while (v < plen && h < tlen && q[v] == t[h]) {
            v++;
            h++;
            run++;
        }
        return run;
    };
    auto start_run = [&](int op, int q0, int t0) {
        if (cur_op != op) {
            flush();
            cur_op = op;
            cur_n = 0;
            run_q = q0;
            run_t = t0;
        }
    };
    {
        const int r = extend();
        if (r > 0) {
            cur_op = 'M';
            cur_n = r;
            run_q = 0;
            run_t = 0;
        }
    }
    for (size_t i = rev.size(); i-- > 0;) {
        const int ob = rev[i], op = ob & 3;
        if (op == 0) {
            start_run('X', v, h);
            cur_n++;
            v++;
            h++;
        } else if (op == 1) {
            start_run('I', v, h);
            cur_n++;
            h++;
        } else {
            start_run('D', v, h);
            cur_n++;
            v++;
        }
        if (ob & 4) {
            const int q1 = v, t1 = h;
            const int r = extend();
            if (r > 0) {
                flush();
                cur_op = 'M';
                cur_n = r;
                run_q = q1;
                run_t = t1;
            }
        }
    }
    flush();
    o->status = 0;
    o->score = s_final;
    o->nops = wp;
    o->qbegin = qbegin;
    o->tbegin = tbegin;
    o->qend = qend;
    o->tend = tend;
    o->align_len = (uint32_t)c_alen;
    o->matches = (uint32_t)c_matches;
    o->gaps = (uint32_t)c_gaps;
    o->gap_regions = (uint32_t)c_greg;
    if (overflow || v != plen || h != tlen) return -1;
    if (!seen_m) o->status = 2;
    return 0;
}

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
