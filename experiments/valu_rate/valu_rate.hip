// valu_rate.hip - how many cycles does a SIMD of gfx950 need per wave64 VALU instruction of the kinds the WFA kernels are
// made of (32-bit integer add / compare+select / DPP move / packed 16-bit min / 64-bit shift, f32 FMA for reference), and
// per SALU instruction per CU ?  Every wave runs ITER x 64 independent instructions of one kind; 1, 4 or 8 waves per SIMD.
// Prints instructions per second per SIMD (per CU for the scalar kind) and cycles at the measured time, assuming 2.4 GHz.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define ITER 4000
#define REP8(x) x x x x x x x x

template <int KIND> __global__ __launch_bounds__(64) void k_rate(uint32_t *out, uint32_t seed) {
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    uint32_t b = seed | 1;
    float f0 = a0, f1 = a1, f2 = a2, f3 = a3, f4 = a4, f5 = a5, f6 = a6, f7 = a7, g = 1.0001f;
    uint64_t q0 = a0, q1 = a1, q2 = a2, q3 = a3;
    uint32_t s0 = seed, s1 = seed * 3, s2 = seed * 5, s3 = seed * 7;
    for (int i = 0; i < ITER; i++) {
        if (KIND == 0) { // v_add_u32
            REP8(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                              "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        } else if (KIND == 1) { // v_fma_f32
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                              "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8"
                              : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(g));)
        } else if (KIND == 2) { // v_cmp + v_cndmask pairs (64 instructions = 32 pairs)
            REP8(asm volatile("v_cmp_lt_u32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_lt_u32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %2, vcc\n"
                              "v_cmp_lt_u32 vcc, %2, %4\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_u32 vcc, %3, %4\n v_cndmask_b32 %3, %3, %0, vcc"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");)
        } else if (KIND == 3) { // v_mov_b32 with DPP (row_ror:4) + v_min_i32: the reduction step
            REP8(asm volatile("s_nop 1\n v_mov_b32_dpp %4, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n v_min_i32 %0, %4, %0\n"
                              "v_mov_b32_dpp %5, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n v_min_i32 %1, %5, %1\n"
                              "v_mov_b32_dpp %6, %2 row_ror:4 row_mask:0xf bank_mask:0xf\n v_min_i32 %2, %6, %2\n"
                              "v_mov_b32_dpp %7, %3 row_ror:4 row_mask:0xf bank_mask:0xf\n v_min_i32 %3, %7, %3"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 4) { // v_pk_min_u16
            REP8(asm volatile("v_pk_min_u16 %0, %0, %8\n v_pk_min_u16 %1, %1, %8\n v_pk_min_u16 %2, %2, %8\n v_pk_min_u16 %3, %3, %8\n"
                              "v_pk_min_u16 %4, %4, %8\n v_pk_min_u16 %5, %5, %8\n v_pk_min_u16 %6, %6, %8\n v_pk_min_u16 %7, %7, %8"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        } else if (KIND == 5) { // v_lshlrev_b64
            REP8(asm volatile("v_lshlrev_b64 %0, 1, %0\n v_lshlrev_b64 %1, 1, %1\n v_lshlrev_b64 %2, 1, %2\n v_lshlrev_b64 %3, 1, %3\n"
                              "v_lshlrev_b64 %0, 1, %0\n v_lshlrev_b64 %1, 1, %1\n v_lshlrev_b64 %2, 1, %2\n v_lshlrev_b64 %3, 1, %3"
                              : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));)
        } else { // s_add_u32: the scalar unit (one per CU)
            REP8(asm volatile("s_add_u32 %0, %0, %1\n s_add_u32 %1, %1, %2\n s_add_u32 %2, %2, %3\n s_add_u32 %3, %3, %0\n"
                              "s_add_u32 %0, %0, %1\n s_add_u32 %1, %1, %2\n s_add_u32 %2, %2, %3\n s_add_u32 %3, %3, %0"
                              : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)::"scc");)
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7) ^ (uint32_t)(q0 ^ q1 ^ q2 ^ q3) ^
                                          s0 ^ s1 ^ s2 ^ s3;
}

template <int KIND> static void run(const char *name, uint32_t *out, int cus) {
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = cus * 4 * wps;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(64), 0, 0, out, 1u);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(64), 0, 0, out, 2u);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double per_wave = (double)ITER * 64;
        const double cyc = ms * 1e-3 * 2.4e9;
        printf("{\"kind\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"cycles_per_instr_per_wave\": %.3f, \"cycles_per_instr_per_simd\": %.3f, "
               "\"cycles_per_instr_per_cu\": %.3f}\n",
               name, wps, ms, cyc / per_wave, cyc / (per_wave * wps), cyc / (per_wave * wps * 4));
    }
}

int main() {
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    uint32_t *out = nullptr;
    hipMalloc(&out, (size_t)cus * 4 * 8 * 64 * 4);
    run<0>("v_add_u32", out, cus);
    run<1>("v_fma_f32", out, cus);
    run<2>("v_cmp+v_cndmask", out, cus);
    run<3>("v_mov_dpp+v_min (s_nop 1 per 8)", out, cus);
    run<4>("v_pk_min_u16", out, cus);
    run<5>("v_lshlrev_b64", out, cus);
    run<6>("s_add_u32", out, cus);
    hipFree(out);
    return 0;
}
