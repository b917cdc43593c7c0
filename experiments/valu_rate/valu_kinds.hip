// valu_kinds.hip - the issue rate of individual instruction forms (8 waves per SIMD, all CUs): which of the instructions the
// WFA kernels use run at the SIMD-32 rate (~2 cycles per wave64 instruction) and which at half of it.  Every wave runs
// ITER x 64 instructions of one form on 8 independent registers.  Cycles assume 2.4 GHz.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define ITER 2000
#define REP8(x) x x x x x x x x
#define OP8(fmt) fmt("%0") "\n" fmt("%1") "\n" fmt("%2") "\n" fmt("%3") "\n" fmt("%4") "\n" fmt("%5") "\n" fmt("%6") "\n" fmt("%7")

#define K_MAX(r) "v_max_i32 " r ", " r ", %8"
#define K_SUB(r) "v_sub_u32 " r ", " r ", %8"
#define K_AND(r) "v_and_b32 " r ", " r ", %8"
#define K_SHL(r) "v_lshlrev_b32 " r ", 1, " r
#define K_MOV(r) "v_mov_b32 " r ", %8"
#define K_CMP32(r) "v_cmp_lt_u32 vcc, " r ", %8"
#define K_CND32(r) "v_cndmask_b32 " r ", " r ", %8, vcc"
#define K_CMP64(r) "v_cmp_lt_u32 s[20:21], " r ", %8"
#define K_CND64(r) "v_cndmask_b32 " r ", " r ", %8, s[20:21]"
#define K_LSHLADD(r) "v_lshl_add_u32 " r ", " r ", 1, %8"
#define K_ADD3(r) "v_add3_u32 " r ", " r ", %8, %8"
#define K_MINU(r) "v_min_u32 " r ", " r ", %8"
#define K_DPP(r) "v_mov_b32_dpp " r ", %8 row_ror:4 row_mask:0xf bank_mask:0xf"
#define K_XOR(r) "v_xor_b32 " r ", " r ", %8"
#define K_FMAC(r) "v_fmac_f32 " r ", %8, %8"
#define K_ADDS(r) "v_add_u32 " r ", s22, " r
#define K_SUBREV(r) "v_subrev_u32 " r ", s22, " r
#define K_CMPS(r) "v_cmp_lt_u32 vcc, s22, " r
#define K_BFE(r) "v_bfe_u32 " r ", " r ", 1, 8"
#define K_LSHR64(r) "v_alignbit_b32 " r ", " r ", %8, 3"

#define KERNEL(NAME, FMT, ...)                                                                                                      \
    __global__ __launch_bounds__(64) void NAME(uint32_t *out, uint32_t seed) {                                                      \
        uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
        uint32_t b = seed | 1;                                                                                                       \
        asm volatile("s_mov_b32 s22, 5\n s_mov_b64 s[20:21], -1" ::: "s20", "s21", "s22");                                           \
        for (int i = 0; i < ITER; i++) {                                                                                             \
            REP8(asm volatile(OP8(FMT) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : __VA_ARGS__);) \
        }                                                                                                                            \
        out[blockIdx.x * 64 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                                                 \
    }
KERNEL(k_max, K_MAX, "memory")
KERNEL(k_sub, K_SUB, "memory")
KERNEL(k_and, K_AND, "memory")
KERNEL(k_shl, K_SHL, "memory")
KERNEL(k_mov, K_MOV, "memory")
KERNEL(k_cmp32, K_CMP32, "vcc")
KERNEL(k_cnd32, K_CND32, "vcc")
KERNEL(k_cmp64, K_CMP64, "s20", "s21")
KERNEL(k_cnd64, K_CND64, "s20", "s21")
KERNEL(k_lshladd, K_LSHLADD, "memory")
KERNEL(k_add3, K_ADD3, "memory")
KERNEL(k_minu, K_MINU, "memory")
KERNEL(k_dpp, K_DPP, "memory")
KERNEL(k_xor, K_XOR, "memory")
KERNEL(k_fmac, K_FMAC, "memory")
KERNEL(k_adds, K_ADDS, "s22")
KERNEL(k_subrev, K_SUBREV, "s22")
KERNEL(k_cmps, K_CMPS, "vcc", "s22")
KERNEL(k_bfe, K_BFE, "memory")
KERNEL(k_alignbit, K_LSHR64, "memory")

typedef void (*Fn)(uint32_t *, uint32_t);
static void run(const char *name, Fn f, uint32_t *out, int cus) {
    const int blocks = cus * 4 * 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(f, dim3(blocks), dim3(64), 0, 0, out, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(f, dim3(blocks), dim3(64), 0, 0, out, 2u);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("{\"form\": \"%s\", \"waves_per_simd\": 8, \"ms\": %.4f, \"cycles_per_instr_per_simd\": %.3f}\n", name, ms,
           ms * 1e-3 * 2.4e9 / ((double)ITER * 64 * 8));
}
#define RUN(k, label) run(label, k, out, cus)
int main() {
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    uint32_t *out = nullptr;
    hipMalloc(&out, (size_t)cus * 4 * 8 * 64 * 4);
    RUN(k_mov, "v_mov_b32 v, v");
    RUN(k_xor, "v_xor_b32 (VOP2)");
    RUN(k_and, "v_and_b32 (VOP2)");
    RUN(k_sub, "v_sub_u32 (VOP2)");
    RUN(k_max, "v_max_i32 (VOP2)");
    RUN(k_minu, "v_min_u32 (VOP2)");
    RUN(k_shl, "v_lshlrev_b32 v, 1, v (VOP2)");
    RUN(k_fmac, "v_fmac_f32 (VOP2)");
    RUN(k_adds, "v_add_u32 v, s, v (VOP2, scalar operand)");
    RUN(k_subrev, "v_subrev_u32 v, s, v (VOP2, scalar operand)");
    RUN(k_cmp32, "v_cmp_lt_u32 vcc, v, v (VOPC)");
    RUN(k_cmps, "v_cmp_lt_u32 vcc, s, v (VOPC, scalar operand)");
    RUN(k_cnd32, "v_cndmask_b32 v, v, v, vcc (VOP2)");
    RUN(k_cmp64, "v_cmp_lt_u32 s[20:21], v, v (VOP3)");
    RUN(k_cnd64, "v_cndmask_b32 v, v, v, s[20:21] (VOP3)");
    RUN(k_lshladd, "v_lshl_add_u32 (VOP3)");
    RUN(k_add3, "v_add3_u32 (VOP3)");
    RUN(k_bfe, "v_bfe_u32 (VOP3)");
    RUN(k_alignbit, "v_alignbit_b32 (VOP3)");
    RUN(k_dpp, "v_mov_b32 DPP row_ror:4");
    hipFree(out);
    return 0;
}
