// self-test of simt_emu.h: results of the cross-lane operations, and (argv[1] = "diverge" / "early") that lanes which do not
// reach the same operation together stop the run instead of producing a value
#include <string.h>

#include "simt_emu.h"

int main(int argc, char **argv) {
    const bool diverge = argc > 1 && !strcmp(argv[1], "diverge"), early = argc > 1 && !strcmp(argv[1], "early");
    const bool stuck = argc > 1 && !strcmp(argv[1], "stuck");
    static uint32_t lds[64];
    int bad = 0;
    const long n = simt::run_wave([&](int lane) {
        const uint64_t odd = simt::ballot(lane & 1, 1);
        if (odd != 0xaaaaaaaaaaaaaaaaull) bad++;
        if (simt::shfl(lane * 3u, (lane + 5) & 63, 2) != (uint32_t)(((lane + 5) & 63) * 3)) bad++;
        const uint32_t m = simt::row_reduce(100u - lane, 3, [](uint32_t a, uint32_t b) { return a < b ? a : b; });
        if (m != 100u - ((lane & 48) + 15)) bad++;
        lds[lane] = lane * 7;
        simt::barrier(4);
        if (lds[63 - lane] != (uint32_t)(63 - lane) * 7) bad++;
        for (int i = 0; i < 1000; i++) // many generations: the double buffer is reused correctly
            if (simt::shfl((uint32_t)(lane + i), lane ^ 1, 5) != (uint32_t)((lane ^ 1) + i)) bad++;
        if (diverge) {
            if (lane < 32)
                simt::ballot(true, 10);
            else
                simt::ballot(true, 11);
        }
        if (early && lane == 7) return;
        simt::barrier(6);
    });
    // a workgroup of four wavefronts: wave operations are per wavefront, the barrier is for all; a wavefront that waits at
    // the barrier while another one waits for a lane that is at the barrier too is reported, not spun on for ever
    static uint32_t part[4];
    const long n4 = simt::run_block(4, [&](int t) {
        const int wv = simt::wave(), ln = simt::lane();
        if (simt::tid() != t || wv != t / 64 || ln != t % 64) bad++;
        const uint64_t m = simt::ballot(ln < wv + 1, 20); // differs per wavefront
        if (m != (2ull << wv) - 1) bad++;
        const uint32_t mx = simt::wave_reduce((uint32_t)t, 21, [](uint32_t a, uint32_t b) { return a > b ? a : b; });
        if (mx != (uint32_t)(64 * wv + 63)) bad++;
        if (ln == 0) part[wv] = mx;
        for (int i = 0; i < wv * 3; i++) simt::wave_sync(22); // wavefronts run different numbers of wave operations
        simt::barrier(23);
        if (part[0] + part[1] + part[2] + part[3] != 63 + 127 + 191 + 255) bad++;
        if (stuck) {
            if (ln == 5 && wv == 2)
                simt::barrier(30); // one lane at the barrier, its wavefront in a wave operation
            else
                simt::wave_sync(31);
        }
        simt::barrier(24);
    });
    printf("%s collectives=%ld+%ld\n", bad ? "FAILED" : "ok", n, n4);
    return bad ? 1 : 0;
}
