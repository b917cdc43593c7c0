// experiments/pa_chain_pipe/compile_check.hip - k_pa_chain_pipe compiled for gfx950 beside the product's kernels (included as
// text), with a stand-in for the function tools/adopt_pa_chain_pipe.py factors out of k_pa_chain_wave.  Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../lexicmap_amd/csrc -I../../include -c compile_check.hip
#include "../../lexicmap_amd/csrc/lm_kernels.hip"

namespace lm {
__device__ int lm_chain2_backtrack(const LmSub *, int, const LmChain2Opt &, uint64_t *, long long, int, int32_t *, LmChain2 *) { return 0; }
#include "../pa_chain_bt/lm_pa_chain_bt.h"
#include "lm_pa_chain_pipe.h"
} // namespace lm
