#!/usr/bin/env python3
"""experiments/pa_chain/make_integrated.py - the product's k_pa_chain_wave with its DP loop replaced by pa_chain_dp_ring, in a
scratch copy of the sources (experiments/csrc_pa, library experiments/lib_pa/liblexicmap_hip.so); writes
integrate_pa_chain.patch (`git apply` from the root; then add lm_pa_chain_dp.h lm_pa_chain_dp_core.h to the Makefile's
lm_kernels.o rule).  NOT yet run on a GPU: checked on the host SIMT emulator only (tests/test_pa_chain_emulated_cpu.py)."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SRC = os.path.join(ROOT, "lexicmap_amd", "csrc")
DST = os.path.join(ROOT, "experiments", "csrc_pa")


def main():
    shutil.rmtree(DST, ignore_errors=True)
    os.makedirs(DST)
    for f in os.listdir(SRC):
        if f.endswith((".hip", ".h", ".cpp")) or f == "Makefile":
            shutil.copy(os.path.join(SRC, f), os.path.join(DST, f))
    shutil.copy(os.path.join(HERE, "pa_chain_dp.h"), os.path.join(DST, "lm_pa_chain_dp_core.h"))
    shutil.copy(os.path.join(HERE, "lm_pa_chain_dp.h"), os.path.join(DST, "lm_pa_chain_dp.h"))
    k = os.path.join(DST, "lm_kernels.hip")
    s = open(k).read()
    a = s.index("__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {")
    s = s[:a] + '#include "lm_pa_chain_dp.h"\n\n' + s[a:]
    b0 = s.index("        // ---- banded DP (lib-chaining2.go:222-307), candidates j scanned 64 at a time from i-1 downwards ----")
    b1 = s.index("        __syncthreads();\n        // ---- backtrack with the explicit region stack (lane 0), identical to lm_run_chain2's second half ----")
    s = s[:b0] + ("        // ---- banded DP (lib-chaining2.go:222-307): the recent anchors and their scores in an LDS ring (lm_pa_chain_dp_core.h) ----\n"
                  "        long long M = 0;\n        int Mi = 0;\n        pa_chain_dp_ring(a_, n, opt, msi, &pcd_lds, &M, &Mi);\n        __threadfence_block();\n") + s[b1:]
    s = s.replace("    const int lane = threadIdx.x;\n    for (int64_t ti = blockIdx.x; ti < ntasks; ti += gridDim.x) {\n        const int64_t o = pa_off[ti];\n        int n = (int)(pa_off[ti + 1] - o);\n        __syncthreads();",
                  "    const int lane = threadIdx.x;\n    __shared__ PcdLds pcd_lds;\n    for (int64_t ti = blockIdx.x; ti < ntasks; ti += gridDim.x) {\n        const int64_t o = pa_off[ti];\n        int n = (int)(pa_off[ti + 1] - o);\n        __syncthreads();", 1)
    assert "pcd_lds" in s
    open(k, "w").write(s)
    m = os.path.join(DST, "Makefile")
    ms = open(m).read().replace("OUT = ../liblexicmap_hip.so", "OUT = ../lib_pa/liblexicmap_hip.so").replace(
        "lm_kernels.o: lm_kernels.hip lm_kernels.h lm_algos.h", "lm_kernels.o: lm_kernels.hip lm_kernels.h lm_algos.h lm_pa_chain_dp.h lm_pa_chain_dp_core.h")
    open(m, "w").write(ms)
    patch = ""
    for f in sorted(os.listdir(DST)):
        if not f.endswith((".hip", ".h", ".cpp")):
            continue
        old = os.path.join(SRC, f)
        rel = "lexicmap_amd/csrc/" + f
        r = subprocess.run(["diff", "-u", "--label", "a/" + rel if os.path.exists(old) else "/dev/null", "--label", "b/" + rel,
                            old if os.path.exists(old) else "/dev/null", os.path.join(DST, f)], capture_output=True, text=True)
        if r.stdout:
            patch += "diff --git a/%s b/%s\n" % (rel, rel) + ("new file mode 100644\n" if not os.path.exists(old) else "") + r.stdout
    open(os.path.join(HERE, "integrate_pa_chain.patch"), "w").write(patch)
    print("patch: %d lines" % patch.count("\n"))
    if "--no-build" not in sys.argv:
        os.makedirs(os.path.join(ROOT, "experiments", "lib_pa"), exist_ok=True)
        subprocess.check_call(["make", "-s", "-C", DST, "lm_kernels.o"])
        print("lm_kernels.o compiles")


if __name__ == "__main__":
    main()
