// wfa_host_walk.h - plain serial statement of bt_walk + bt_replay (lm_kernels.hip) for the emulator harnesses: edit operations
// from the backtrace rows a forward pass wrote (same byte format as k_wfa_lean's), match runs by greedy extension, run list
// and statistics of the M-trimmed alignment.  Test infrastructure.
#pragma once
#include <stdint.h>

#include <vector>

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

