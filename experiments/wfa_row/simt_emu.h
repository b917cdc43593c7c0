// simt_emu.h - a one-wavefront SIMT emulator for the host: the 64 lanes of a wavefront run as 64 cooperative fibers
// (ucontext), every cross-lane operation (ballot, shuffle, row reduction, LDS hand-off) is a rendezvous of ALL 64 lanes at
// the SAME call site.  A lane that reaches a different cross-lane operation than the others, or that returns while others
// wait, aborts the run with both sites: the kernels written against this layer keep every cross-lane operation in
// wave-uniform control flow, which is also the only form whose hardware semantics do not depend on what inactive lanes
// return.  Test infrastructure (CPU check of experiments/wfa_row/wfa_row_fwd.h before it ever sees a GPU); not the product.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <ucontext.h>

#include <functional>

namespace simt {

struct Wave {
    static constexpr int N = 64;
    ucontext_t main_ctx, ctx[N];
    char *stacks = nullptr;
    bool done[N];
    int cur = 0;
    int ndone = 0;
    // rendezvous state, double buffered by generation parity
    uint64_t xbuf[2][N];
    int count[2] = {0, 0}, readers[2] = {0, 0}, site[2] = {0, 0};
    long gen[N];
    long collectives = 0;
    std::function<void(int)> body;
};

inline Wave *&current() {
    static Wave *w = nullptr;
    return w;
}
inline int lane() { return current()->cur; }

inline void fail(const char *what, int a, int b) {
    fprintf(stderr, "simt_emu: %s (lane %d: site %d vs %d)\n", what, current() ? current()->cur : -1, a, b);
    abort();
}

inline void yield_next() {
    Wave *w = current();
    const int me = w->cur;
    for (int i = 1; i <= Wave::N; i++) {
        const int n = (me + i) % Wave::N;
        if (!w->done[n]) {
            if (n == me) return;
            w->cur = n;
            swapcontext(&w->ctx[me], &w->ctx[n]);
            return;
        }
    }
}

// every lane contributes `v`; returns the buffer holding all 64 contributions once every lane has arrived
inline const uint64_t *rendezvous(uint64_t v, int site_id) {
    Wave *w = current();
    const int me = w->cur;
    const int p = (int)(w->gen[me]++ & 1);
    if (w->ndone) fail("a lane returned while others still run cross-lane operations", site_id, -1);
    if (w->count[p] == 0)
        w->site[p] = site_id;
    else if (w->site[p] != site_id)
        fail("divergent cross-lane operation", site_id, w->site[p]);
    w->xbuf[p][me] = v;
    w->count[p]++;
    if (me == 0) w->collectives++;
    while (w->count[p] < Wave::N) {
        if (w->ndone) fail("a lane returned while others wait", site_id, -1);
        yield_next();
        w->cur = me;
    }
    return w->xbuf[p];
}
inline void leave(int p_gen_parity) {
    Wave *w = current();
    if (++w->readers[p_gen_parity] == Wave::N) {
        w->readers[p_gen_parity] = 0;
        w->count[p_gen_parity] = 0;
    }
}
#define SIMT_PARITY() ((int)((simt::current()->gen[simt::current()->cur] - 1) & 1))

inline uint64_t ballot(bool p, int site_id) {
    const uint64_t *b = rendezvous(p ? 1 : 0, site_id);
    uint64_t m = 0;
    for (int i = 0; i < Wave::N; i++) m |= (b[i] & 1) << i;
    leave(SIMT_PARITY());
    return m;
}
inline uint32_t shfl(uint32_t v, int src, int site_id) {
    const uint64_t *b = rendezvous(v, site_id);
    const uint32_t r = (uint32_t)b[src & 63];
    leave(SIMT_PARITY());
    return r;
}
inline void barrier(int site_id) {
    rendezvous(0, site_id);
    leave(SIMT_PARITY());
}
// reduction over the 16 lanes of the caller's row with `op`
template <typename F> inline uint32_t row_reduce(uint32_t v, int site_id, F op) {
    const uint64_t *b = rendezvous(v, site_id);
    const int r0 = lane() & 48;
    uint32_t acc = (uint32_t)b[r0];
    for (int i = 1; i < 16; i++) acc = op(acc, (uint32_t)b[r0 + i]);
    leave(SIMT_PARITY());
    return acc;
}

static void trampoline(int lane_id) {
    Wave *w = current();
    w->body(lane_id);
    w->done[lane_id] = true;
    w->ndone++;
    for (int p = 0; p < 2; p++)
        if (w->count[p] != 0 && w->count[p] < Wave::N) fail("a lane returned while others wait in a cross-lane operation", w->site[p], -1);
    // hand over to the next unfinished lane, or back to the launcher
    for (int i = 1; i <= Wave::N; i++) {
        const int n = (lane_id + i) % Wave::N;
        if (!w->done[n]) {
            w->cur = n;
            setcontext(&w->ctx[n]);
        }
    }
    setcontext(&w->main_ctx);
}

// runs body(lane) for the 64 lanes of one wavefront; returns the number of cross-lane operations executed
inline long run_wave(std::function<void(int)> body, size_t stack_bytes = 1 << 20) {
    Wave *w = new Wave();
    w->body = body;
    w->stacks = (char *)malloc(stack_bytes * Wave::N);
    current() = w;
    for (int i = 0; i < Wave::N; i++) {
        w->done[i] = false;
        w->gen[i] = 0;
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

} // namespace simt
