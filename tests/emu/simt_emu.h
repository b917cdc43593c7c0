// simt_emu.h - a one-workgroup SIMT emulator for the host: the lanes of 1..16 wavefronts run as cooperative fibers
// (ucontext); every cross-lane operation of a wavefront (ballot, shuffle, reduction, wave-level LDS hand-off) is a rendezvous
// of ALL 64 lanes of that wavefront at the SAME call site, a workgroup barrier a rendezvous of all lanes of the workgroup.
// A lane that reaches a different operation than the others of its group, or that returns while others wait, or a set of
// lanes that can no longer make progress (some at a barrier, others in a wave operation) aborts the run with the sites: the
// kernels written against this layer keep every cross-lane operation in uniform control flow of its group, which is also the
// only form whose hardware semantics do not depend on what inactive lanes return.  Between rendezvous a lane runs alone, in
// any order relative to the others: an LDS or global hand-off that is not separated by a rendezvous shows up as a wrong result.
// Test infrastructure (CPU check of experiments/wfa_row/*_fwd.h before they ever see a GPU); not the product.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <ucontext.h>

#include <functional>
#include <vector>

namespace simt {

struct Group { // a wavefront, or the whole workgroup
    int first = 0, size = 0;
    std::vector<uint64_t> xbuf[2];
    int count[2] = {0, 0}, readers[2] = {0, 0}, site[2] = {0, 0};
};

struct Block {
    int nlanes = 64, nwaves = 1;
    ucontext_t main_ctx;
    std::vector<ucontext_t> ctx;
    char *stacks = nullptr;
    std::vector<char> done;
    int cur = 0, ndone = 0;
    std::vector<Group> groups;           // [0, nwaves): the wavefronts; [nwaves]: the workgroup
    std::vector<long> gen_wave, gen_blk; // per lane: operations of its wavefront / of the workgroup so far
    long collectives = 0, spin = 0;
    std::function<void(int)> body;
};

inline Block *&current() {
    static Block *b = nullptr;
    return b;
}
inline int tid() { return current()->cur; }
inline int lane() { return current()->cur & 63; }
inline int wave() { return current()->cur >> 6; }

inline void fail(const char *what, int a, int b) {
    fprintf(stderr, "simt_emu: %s (thread %d: site %d vs %d)\n", what, current() ? current()->cur : -1, a, b);
    abort();
}

inline void yield_next() {
    Block *w = current();
    const int me = w->cur;
    for (int i = 1; i <= w->nlanes; i++) {
        const int n = (me + i) % w->nlanes;
        if (!w->done[n]) {
            if (n == me) return;
            w->cur = n;
            swapcontext(&w->ctx[me], &w->ctx[n]);
            return;
        }
    }
}

// every lane of the group contributes `v`; returns the buffer holding all contributions once every lane has arrived
// (index = lane - first lane of the group), and the parity to hand to leave()
inline const uint64_t *rendezvous(bool whole_block, uint64_t v, int site_id, int *parity) {
    Block *w = current();
    const int me = w->cur;
    Group &g = w->groups[whole_block ? w->nwaves : (me >> 6)];
    long &gen = whole_block ? w->gen_blk[me] : w->gen_wave[me];
    const int p = (int)(gen++ & 1);
    *parity = p;
    if (w->ndone) fail("a lane returned while others still run cross-lane operations", site_id, -1);
    if (g.count[p] == 0)
        g.site[p] = site_id;
    else if (g.site[p] != site_id)
        fail("divergent cross-lane operation", site_id, g.site[p]);
    g.xbuf[p][me - g.first] = v;
    g.count[p]++;
    w->spin = 0;
    if (me == g.first) w->collectives++;
    while (g.count[p] < g.size) {
        if (w->ndone) fail("a lane returned while others wait", site_id, -1);
        if (++w->spin > 8L * w->nlanes) fail("no lane can make progress (lanes of one group wait at different kinds of operation)", site_id, -1);
        yield_next();
        w->cur = me;
    }
    return g.xbuf[p].data();
}
inline void leave(bool whole_block, int p) {
    Block *w = current();
    Group &g = w->groups[whole_block ? w->nwaves : (w->cur >> 6)];
    if (++g.readers[p] == g.size) {
        g.readers[p] = 0;
        g.count[p] = 0;
    }
}

inline uint64_t ballot(bool pr, int site_id) { // of the caller's wavefront
    int p;
    const uint64_t *b = rendezvous(false, pr ? 1 : 0, site_id, &p);
    uint64_t m = 0;
    for (int i = 0; i < 64; i++) m |= (b[i] & 1) << i;
    leave(false, p);
    return m;
}
inline uint32_t shfl(uint32_t v, int src, int site_id) {
    int p;
    const uint64_t *b = rendezvous(false, v, site_id, &p);
    const uint32_t r = (uint32_t)b[src & 63];
    leave(false, p);
    return r;
}
inline void wave_sync(int site_id) { // hand-off through LDS inside one wavefront
    int p;
    rendezvous(false, 0, site_id, &p);
    leave(false, p);
}
inline void barrier(int site_id) { // of the workgroup
    int p;
    rendezvous(true, 0, site_id, &p);
    leave(true, p);
}
// reduction over the 16 lanes of the caller's row / the 64 lanes of its wavefront with `op`
template <typename F> inline uint32_t row_reduce(uint32_t v, int site_id, F op) {
    int p;
    const uint64_t *b = rendezvous(false, v, site_id, &p);
    const int r0 = lane() & 48;
    uint32_t acc = (uint32_t)b[r0];
    for (int i = 1; i < 16; i++) acc = op(acc, (uint32_t)b[r0 + i]);
    leave(false, p);
    return acc;
}
template <typename F> inline uint32_t wave_reduce(uint32_t v, int site_id, F op) {
    int p;
    const uint64_t *b = rendezvous(false, v, site_id, &p);
    uint32_t acc = (uint32_t)b[0];
    for (int i = 1; i < 64; i++) acc = op(acc, (uint32_t)b[i]);
    leave(false, p);
    return acc;
}

static void trampoline(int id) {
    Block *w = current();
    w->body(id);
    w->done[id] = 1;
    w->ndone++;
    for (auto &g : w->groups)
        for (int p = 0; p < 2; p++)
            if (g.count[p] != 0 && g.count[p] < g.size) fail("a lane returned while others wait in a cross-lane operation", g.site[p], -1);
    for (int i = 1; i <= w->nlanes; i++) { // hand over to the next unfinished lane, or back to the launcher
        const int n = (id + i) % w->nlanes;
        if (!w->done[n]) {
            w->cur = n;
            setcontext(&w->ctx[n]);
        }
    }
    setcontext(&w->main_ctx);
}

// runs body(thread) for the 64 * nwaves threads of one workgroup; returns the number of cross-lane operations executed
inline long run_block(int nwaves, std::function<void(int)> body, size_t stack_bytes = 1 << 20) {
    Block *w = new Block();
    w->nwaves = nwaves;
    w->nlanes = 64 * nwaves;
    w->body = body;
    w->stacks = (char *)malloc(stack_bytes * w->nlanes);
    w->ctx.resize(w->nlanes);
    w->done.assign(w->nlanes, 0);
    w->gen_wave.assign(w->nlanes, 0);
    w->gen_blk.assign(w->nlanes, 0);
    w->groups.resize(nwaves + 1);
    for (int g = 0; g <= nwaves; g++) {
        w->groups[g].first = g < nwaves ? 64 * g : 0;
        w->groups[g].size = g < nwaves ? 64 : w->nlanes;
        w->groups[g].xbuf[0].assign(w->groups[g].size, 0);
        w->groups[g].xbuf[1].assign(w->groups[g].size, 0);
    }
    current() = w;
    for (int i = 0; i < w->nlanes; i++) {
        getcontext(&w->ctx[i]);
        w->ctx[i].uc_stack.ss_sp = w->stacks + stack_bytes * i;
        w->ctx[i].uc_stack.ss_size = stack_bytes;
        w->ctx[i].uc_link = nullptr;
        makecontext(&w->ctx[i], (void (*)())trampoline, 1, i);
    }
    w->cur = 0;
    swapcontext(&w->main_ctx, &w->ctx[0]);
    const long n = w->collectives;
    free(w->stacks);
    current() = nullptr;
    delete w;
    return n;
}
inline long run_wave(std::function<void(int)> body, size_t stack_bytes = 1 << 20) { return run_block(1, body, stack_bytes); }

} // namespace simt
