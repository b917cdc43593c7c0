#!/usr/bin/env python3
"""adopt_pa_chain_pipe.py [--root DIR]: puts the staged k_pa_chain_pipe (experiments/pa_chain_pipe: the Chainer2 DP of a long
chaining window by a workgroup of eight pipelined wavefronts) into the product sources under DIR (default: this repository).
k_pa_chain_wave keeps unpack / ClearSubstrPairs / Trim of every window and the short windows whole; windows with more than
LM_PA_PIPE_MIN (512) anchors after the trim are handed to k_pa_chain_pipe; switch LM_PA_CHAIN_PIPE (default on).  Also the backtrack by the wavefront
(experiments/pa_chain_bt: region scans by 64 lanes, the walk out of LDS tiles) in both kernels, switch LM_PA_CHAIN_BT_WAVE.  The backtrack
block of k_pa_chain_wave becomes the function lm_chain2_backtrack both kernels call.  Every edit is asserted against the text
it replaces.  (Checked in round 4 on a copy of the tree: the library builds.)  Round 5: run it, build, then
tests/test_gpu_parity.py (pseudo-alignment chains), tests/test_gpu_c4c5.py, tests/test_gpu_longreads.py with the switch on
and off, then a C4 shard line."""
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) >= 3 and sys.argv[1] == "--root":
    root = os.path.abspath(sys.argv[2])
src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "experiments", "pa_chain_pipe")
csrc = os.path.join(root, "lexicmap_amd", "csrc")


def edit(path, pairs):
    s = open(path).read()
    for old, new in pairs:
        assert s.count(old) == 1, (path, old[:70], s.count(old))
        s = s.replace(old, new)
    open(path, "w").write(s)


for f, g in (("pa_chain_pipe.h", "lm_pa_chain_pipe_dp.h"), ("lm_pa_chain_pipe.h", "lm_pa_chain_pipe.h")):
    t = open(os.path.join(src, f)).read().replace("experiments/pa_chain_pipe/" + f, g).replace("(STAGED for round 5)", "")
    open(os.path.join(csrc, g), "w").write(t)
edit(os.path.join(csrc, "lm_pa_chain_pipe.h"), [('#include "pa_chain_pipe.h"', '#include "lm_pa_chain_pipe_dp.h"')])
# the wavefront backtrack (experiments/pa_chain_bt), switch LM_PA_CHAIN_BT_WAVE
src_bt = os.path.join(os.path.dirname(src), "pa_chain_bt")
for f, g in (("pa_chain_bt.h", "lm_pa_chain_bt_core.h"), ("pa_clear_tile.h", "lm_pa_clear_tile.h"), ("lm_pa_chain_bt.h", "lm_pa_chain_bt.h")):
    t = open(os.path.join(src_bt, f)).read().replace("experiments/pa_chain_bt/" + f, g).replace("(STAGED for round 5)", "")
    open(os.path.join(csrc, g), "w").write(t)
edit(os.path.join(csrc, "lm_pa_chain_bt.h"), [('#include "pa_chain_bt.h"', '#include "lm_pa_chain_bt_core.h"'), ('#include "pa_clear_tile.h"', '#include "lm_pa_clear_tile.h"')])

# 1. the backtrack block of k_pa_chain_wave as a function
k = os.path.join(csrc, "lm_kernels.hip")
s = open(k).read()
a = "        // ---- backtrack with the explicit region stack (lane 0), identical to lm_run_chain2's second half ----\n        if (lane == 0) {\n            int nout = 0;\n"
b = "            out_n[ti] = nout;\n        }\n    }\n}\n"
assert s.count(a) == 1 and s.count(b) == 1
i, j = s.index(a), s.index(b)
body = s[i + len(a):j]
decl = "                int32_t *stack = stack_pool + 2 * o + 4 * ti;\n"
assert body.count(decl) == 1
body = body.replace(decl, "")
body = "\n".join(ln[8:] if ln.startswith("        ") else ln for ln in body.split("\n"))  # one level out
fn = ("// Backtrack with the explicit region stack: lm_run_chain2's second half, by one thread (k_pa_chain_wave's lane 0,\n"
      "// k_pa_chain_pipe's thread 0).  msi[]: (score << 32 | predecessor) per anchor, M / Mi the best score and its anchor.\n"
      "__device__ int lm_chain2_backtrack(const LmSub *a_, int n, const LmChain2Opt &opt, const uint64_t *msi, long long M, int Mi, int32_t *stack,\n"
      "                                   LmChain2 *res) {\n    int nout = 0;\n" + body + "    return nout;\n}\n\n")
s = s[:i] + ("        // ---- backtrack (lm_chain2_backtrack), or the hand-over of a long window to k_pa_chain_pipe ----\n"
             "        if (dbg) d_2 = wall_clock64();\n"
             "        if (bt_wave & 1) { // by the wavefront: region scans by 64 lanes, the walk out of LDS tiles (lm_pa_chain_bt.h)\n"
             "            const int no = pa_chain_backtrack_wave(a_, n, opt, msi, M, Mi, stack_pool + 2 * o + 4 * ti, res, &pcb_lds);\n"
             "            if (lane == 0) out_n[ti] = no;\n"
             "        } else if (lane == 0) {\n"
             "            out_n[ti] = lm_chain2_backtrack(a_, n, opt, msi, M, Mi, stack_pool + 2 * o + 4 * ti, res);\n"
             "        }\n"
             "        if (dbg && lane == 0) {\n"
             "            const unsigned long long d_3 = wall_clock64();\n"
             "            atomicAdd(dbg + 0, d_1 - d_0);\n            atomicAdd(dbg + 1, d_2 - d_1);\n            atomicAdd(dbg + 2, d_3 - d_2);\n            atomicAdd(dbg + 3, 1ull);\n"
             "        }\n"
             "    }\n}\n") + s[j + len(b):]
h = "// RING: the DP keeps the recent anchors and scores in an LDS ring (lm_pa_chain_dp_core.h); otherwise every step goes\n"
assert s.count(h) == 1
s = s.replace(h, fn + h)
open(k, "w").write(s)

edit(k, [
    # 2. k_pa_chain_wave hands long windows over
    ("                                                       int32_t *__restrict__ out_n, int32_t *__restrict__ clr_n, int qbits,\n"
     "                                                       int tbits) {\n    const int lane = threadIdx.x;\n    __shared__ PcdLds pcd_lds;\n",
     "                                                       int32_t *__restrict__ out_n, int32_t *__restrict__ clr_n, int qbits,\n"
     "                                                       int tbits, int pipe_min, int32_t *__restrict__ long_tasks,\n"
     "                                                       unsigned int *__restrict__ nlong, int bt_wave,\n"
     "                                                       unsigned long long *__restrict__ dbg) {\n    const int lane = threadIdx.x;\n    __shared__ PcdLds pcd_lds;\n    __shared__ PcbLds pcb_lds;\n"),
    # LM_DEBUG_PA_CHAIN (dbg != nullptr): where a window's time goes - unpack + clear + trim / DP / backtrack, summed over the
    # windows this kernel finishes itself, on the 100-MHz wall clock
    ("        LmChain2 *res = out_pool + o;\n        for (int i = lane; i < n; i += 64) {\n            const uint64_t v = B[o + i];\n",
     "        LmChain2 *res = out_pool + o;\n        unsigned long long d_0 = 0, d_1 = 0, d_2 = 0;\n        if (dbg) d_0 = wall_clock64();\n"
     "        for (int i = lane; i < n; i += 64) {\n            const uint64_t v = B[o + i];\n"),
    ("        long long M = 0;\n        int Mi = 0;\n        if (RING) {\n            pa_chain_dp_ring(", "        if (dbg) d_1 = wall_clock64();\n        long long M = 0;\n        int Mi = 0;\n        if (RING) {\n            pa_chain_dp_ring("),
    ('#include "lm_pa_chain_dp.h"\n', '#include "lm_pa_chain_dp.h"\n#include "lm_pa_chain_bt.h"\n'),
    ("        const LmSub *a_ = sb + start;\n        if (n == 1) {\n",
     "        const LmSub *a_ = sb + start;\n"
     "        if (pipe_min > 0 && n > pipe_min) { // a long window: its DP and backtrack by a workgroup (k_pa_chain_pipe)\n"
     "            if (lane == 0) {\n                out_n[ti] = start;\n                long_tasks[atomicAdd(nlong, 1u)] = (int32_t)ti;\n            }\n"
     "            continue;\n        }\n        if (n == 1) {\n"),
    # 2b. the marks of ClearSubstrPairs from LDS tiles (the tile aliases the DP's ring: the DP comes later)
    ("            for (int i = lane; i < n; i += 64) {\n                uint8_t mk = 0;\n                if (i >= 1) {\n                    const LmSub v = sb[i];\n",
     "            if (bt_wave & 2) pa_clear_marks_wave(sb, n, K, marks, (PccLds *)&pcd_lds);\n"
     "            for (int i = lane; i < n && !(bt_wave & 2); i += 64) {\n                uint8_t mk = 0;\n                if (i >= 1) {\n                    const LmSub v = sb[i];\n"),
    # 3. the workgroup kernel and the launcher
    ("__global__ void k_gather_chain2(", '#include "lm_pa_chain_pipe.h"\n\n__global__ void k_gather_chain2('),
    ("                     int32_t *clr_n, int qbits, int tbits, bool ring) {\n"
     "    int g = (int)(ntasks < 1 ? 1 : (ntasks > 262144 ? 262144 : ntasks));\n"
     "    hipLaunchKernelGGL(ring ? k_pa_chain_wave<true> : k_pa_chain_wave<false>, dim3(g), dim3(64), 0, st, B, pa_off, ntasks, K, opt, subs, marks, msi, stack, out, out_n,\n"
     "                       clr_n, qbits, tbits);\n",
     "                     int32_t *clr_n, int qbits, int tbits, bool ring, int pipe_min, int64_t total, int bt_wave) {\n"
     "    int g = (int)(ntasks < 1 ? 1 : (ntasks > 262144 ? 262144 : ntasks));\n"
     "    // pipe_min > 0: the list of long windows and its counter live behind the stacks (stack holds 2 * total + 5 * ntasks + 48 ints)\n"
     "    int32_t *long_tasks = stack + 2 * total + 4 * ntasks + 8;\n"
     "    unsigned int *nlong = (unsigned int *)(long_tasks + ntasks);\n"
     "    static const bool pa_dbg = getenv(\"LM_DEBUG_PA_CHAIN\") != nullptr; // phase times of k_pa_chain_wave (stack holds 16 more ints)\n"
     "    unsigned long long *dbg = pa_dbg ? (unsigned long long *)(((uintptr_t)(nlong + 2) + 7) & ~(uintptr_t)7) : nullptr;\n"
     "    if (dbg) (void)hipMemsetAsync(dbg, 0, 4 * sizeof(unsigned long long), st);\n"
     "    if (pipe_min > 0) (void)hipMemsetAsync(nlong, 0, sizeof(unsigned int), st);\n"
     "    hipLaunchKernelGGL(ring ? k_pa_chain_wave<true> : k_pa_chain_wave<false>, dim3(g), dim3(64), 0, st, B, pa_off, ntasks, K, opt, subs, marks, msi, stack, out, out_n,\n"
     "                       clr_n, qbits, tbits, pipe_min, long_tasks, nlong, bt_wave, dbg);\n"
     "    if (pipe_min > 0) // (the number of long windows is known on the device only: a grid that fills the chip, workgroups loop)\n"
     "        hipLaunchKernelGGL(k_pa_chain_pipe, dim3((unsigned)(ntasks < 1024 ? (ntasks < 1 ? 1 : ntasks) : 1024)), dim3(PCP_NW * 64), 0, st, pa_off, long_tasks,\n"
     "                           nlong, opt, subs, msi, stack, out, out_n, clr_n, bt_wave & 1);\n"
     "    if (dbg) {\n"
     "        unsigned long long h[4] = {0, 0, 0, 0};\n        unsigned int nl = 0;\n"
     "        (void)hipStreamSynchronize(st);\n        (void)hipMemcpy(h, dbg, sizeof h, hipMemcpyDeviceToHost);\n        (void)hipMemcpy(&nl, nlong, sizeof nl, hipMemcpyDeviceToHost);\n"
     "        fprintf(stderr, \"[lm] k_pa_chain: %lld windows (%llu finished by the wavefront kernel, %u handed to the workgroup kernel), wavefront-ms: clear+trim %.1f, DP %.1f, backtrack %.1f\\n\",\n"
     "                (long long)ntasks, h[3], pipe_min > 0 ? nl : 0u, (double)h[0] / 1e5, (double)h[1] / 1e5, (double)h[2] / 1e5);\n"
     "    }\n"),
])
edit(os.path.join(csrc, "lm_kernels.h"), [
    ("                     int32_t *clr_n, int qbits, int tbits, bool ring);", "                     int32_t *clr_n, int qbits, int tbits, bool ring, int pipe_min = 0, int64_t total = 0, int bt_wave = 0);"),
])
edit(os.path.join(csrc, "lm_internal.h"), [
    ("    int pa_chain_ring = 1;   // Chainer2 DP with the recent anchors in an LDS ring",
     "    int pa_chain_pipe = 1;   // the Chainer2 DP of windows with more than pa_pipe_min anchors by a workgroup of pipelined wavefronts (LM_PA_CHAIN_PIPE=0: off)\n"
     "    int pa_pipe_min = 512;   // LM_PA_PIPE_MIN\n"
     "    int pa_chain_bt_wave = 3; // LM_PA_CHAIN_BT_WAVE: bit 0 = the backtrack of Chainer2 by the wavefront (LDS tiles, 64-lane region scans; 0: lane 0), bit 1 = the marks of ClearSubstrPairs from LDS tiles (0: binary search + scan in global memory)\n"
     "    int pa_chain_ring = 1;   // Chainer2 DP with the recent anchors in an LDS ring"),
    ('        if (const char *e = getenv("LM_PA_CHAIN_RING")) pa_chain_ring = atoi(e) != 0;\n',
     '        if (const char *e = getenv("LM_PA_CHAIN_RING")) pa_chain_ring = atoi(e) != 0;\n'
     '        if (const char *e = getenv("LM_PA_CHAIN_PIPE")) pa_chain_pipe = atoi(e) != 0;\n'
     '        if (const char *e = getenv("LM_PA_PIPE_MIN")) pa_pipe_min = std::max(64, atoi(e));\n'
     '        if (const char *e = getenv("LM_PA_CHAIN_BT_WAVE")) pa_chain_bt_wave = atoi(e) & 3;\n'),
])
edit(os.path.join(csrc, "lm_pipeline.hip"), [
    ("        a.stack.ensure(2 * (size_t)TP + 4 * (size_t)nt + 8);\n", "        a.stack.ensure(2 * (size_t)TP + 5 * (size_t)nt + 48); // (+ the list of long windows, its counter, the debug counters)\n"),
    ("a.out.p, a.out_n.p, a.clr_n.p, compact ? qbits : 0, compact ? tbits : 0, ix->tune.pa_chain_ring != 0);",
     "a.out.p, a.out_n.p, a.clr_n.p, compact ? qbits : 0, compact ? tbits : 0, ix->tune.pa_chain_ring != 0,\n"
     "                            ix->tune.pa_chain_pipe ? ix->tune.pa_pipe_min : 0, TP, ix->tune.pa_chain_bt_wave);"),
])
edit(os.path.join(csrc, "Makefile"), [
    ("lm_pa_chain_dp.h lm_pa_chain_dp_core.h\n", "lm_pa_chain_dp.h lm_pa_chain_dp_core.h lm_pa_chain_pipe.h lm_pa_chain_pipe_dp.h lm_pa_chain_bt.h lm_pa_chain_bt_core.h lm_pa_clear_tile.h\n"),
])
# the rows of the long-read fixture with the new switches off (tests/test_gpu_longreads.py: every pair of device paths agrees)
t = os.path.join(root, "tests", "test_gpu_longreads.py")
if os.path.exists(t):
    edit(t, [('for var, off in (("LM_WFA_MW", "0"), ', 'for var, off in (("LM_WFA_MW", "0"), (\"LM_PA_CHAIN_PIPE\", \"0\"), (\"LM_PA_CHAIN_BT_WAVE\", \"0\"), (\"LM_PA_PIPE_MIN\", \"64\"), ')])
t = os.path.join(root, "tests", "test_adopt_scripts_cpu.py")  # (checks that the scripts apply to the UNadopted tree: done with)
if os.path.exists(t):
    os.remove(t)
print("k_pa_chain_pipe adopted under", root)
