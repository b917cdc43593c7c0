// experiments/wfa_lean2/compile_check.hip - k_wfa_lean2 compiled for gfx950 beside the product's kernels (included as text).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../lexicmap_amd/csrc -I../../include --cuda-device-only -S compile_check.hip
// (tools in this directory read the assembly: isa_loops.py)
#include "../../lexicmap_amd/csrc/lm_kernels.hip"

namespace lm {
#include "lm_wfa_lean2.h"
#include "lm_wfa_mw2.h"
template __global__ void k_wfa_lean2<2, int16_t, false>(const WfaIn *, int64_t, const int32_t *, int64_t, int32_t *, int64_t, uint8_t *, int64_t, uint64_t *,
                                                 unsigned int *, int, int, WfaOut *, unsigned long long *);
template __global__ void k_wfa_lean2<4, int16_t, false>(const WfaIn *, int64_t, const int32_t *, int64_t, int32_t *, int64_t, uint8_t *, int64_t, uint64_t *,
                                                 unsigned int *, int, int, WfaOut *, unsigned long long *);
template __global__ void k_wfa_lean2<2, int32_t, false>(const WfaIn *, int64_t, const int32_t *, int64_t, int32_t *, int64_t, uint8_t *, int64_t, uint64_t *,
                                                 unsigned int *, int, int, WfaOut *, unsigned long long *);
template __global__ void k_wfa_lean2<4, int32_t, true>(const WfaIn *, int64_t, const int32_t *, int64_t, int32_t *, int64_t, uint8_t *, int64_t, uint64_t *,
                                                      unsigned int *, int, int, WfaOut *, unsigned long long *);
template __global__ void k_wfa_lean2<8, int32_t, true>(const WfaIn *, int64_t, const int32_t *, int64_t, int32_t *, int64_t, uint8_t *, int64_t, uint64_t *,
                                                      unsigned int *, int, int, WfaOut *, unsigned long long *);
template __global__ void k_wfa_mw2<2, false>(const WfaIn *, int64_t, const int32_t *, int64_t, int32_t *, int64_t, uint8_t *, int64_t, uint64_t *,
                                             unsigned int *, int, int, WfaOut *);
template __global__ void k_wfa_mw2<4, false>(const WfaIn *, int64_t, const int32_t *, int64_t, int32_t *, int64_t, uint8_t *, int64_t, uint64_t *,
                                             unsigned int *, int, int, WfaOut *);
template __global__ void k_wfa_mw2<2, true>(const WfaIn *, int64_t, const int32_t *, int64_t, int32_t *, int64_t, uint8_t *, int64_t, uint64_t *,
                                            unsigned int *, int, int, WfaOut *);
template __global__ void k_wfa_mw2<4, true>(const WfaIn *, int64_t, const int32_t *, int64_t, int32_t *, int64_t, uint8_t *, int64_t, uint64_t *,
                                            unsigned int *, int, int, WfaOut *);
} // namespace lm
