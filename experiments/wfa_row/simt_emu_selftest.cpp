// self-test of simt_emu.h: results of the cross-lane operations, and (argv[1] = "diverge" / "early") that lanes which do not
// reach the same operation together stop the run instead of producing a value
#include <string.h>

#include "simt_emu.h"

int main(int argc, char **argv) {
    const bool diverge = argc > 1 && !strcmp(argv[1], "diverge"), early = argc > 1 && !strcmp(argv[1], "early");
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
    printf("%s collectives=%ld\n", bad ? "FAILED" : "ok", n);
    return bad ? 1 : 0;
}
