#!/usr/bin/env python3
"""adopt_pa_search_debug.py [--root DIR]: LM_DEBUG_PA_SEARCH - counters and phase clocks of k_pa_search (the second largest
kernel of a C3 step: 4.7 s exclusive for 0.3e12 instructions).  Per launch: candidates searched, wavefront passes, passes of
the match enumeration, anchors staged, wavefront-ms in the search (candidate -> k-mer -> bucket table -> lower bound) and in
the enumeration (keys / values of the matches, staging, flushes) on the 100-MHz wall clock.  No effect without the variable
(a null pointer test per pass).  What DESIGN.md 9b.6 needs before k_pa_search is touched: whether it is bound by the dependent
loads of the search, by the enumeration's near-empty passes, or by the line traffic of both.  Asserted edits; the adopted tree
builds (round 4, on a copy)."""
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) >= 3 and sys.argv[1] == "--root":
    root = os.path.abspath(sys.argv[2])
csrc = os.path.join(root, "lexicmap_amd", "csrc")


def edit(path, pairs):
    s = open(path).read()
    for old, new in pairs:
        assert s.count(old) == 1, (path, old[:70], s.count(old))
        s = s.replace(old, new)
    open(path, "w").write(s)


edit(os.path.join(csrc, "lm_kernels.hip"), [
    ("                                                    int tbits, int nseg, int xcd_map) {\n    // qbits > 0: compact single-key anchors,",
     "                                                    int tbits, int nseg, int xcd_map, unsigned long long *__restrict__ dbg) {\n    // qbits > 0: compact single-key anchors,"),
    ("        const int64_t ci = base + threadIdx.x;\n        int j = 0, hi = 0, i = 0;\n",
     "        const int64_t ci = base + threadIdx.x;\n        unsigned long long d_0 = 0, d_1 = 0, d_it = 0, d_an = 0;\n        if (dbg) d_0 = wall_clock64();\n        int j = 0, hi = 0, i = 0;\n"),
    ("        // the matches of all lanes, one per lane and round, appended to the wavefront's LDS strip\n        while (true) {\n            bool live = j < hi;\n",
     "        // the matches of all lanes, one per lane and round, appended to the wavefront's LDS strip\n        if (dbg) d_1 = wall_clock64();\n        while (true) {\n            d_it++;\n            bool live = j < hi;\n"),
    ("                n_stg += __popcll(m);\n                if (n_stg > PAS_STAGE - 64) flush(); // LDS accesses of one wavefront complete in program order\n            }\n        }\n",
     "                n_stg += __popcll(m);\n                d_an += (unsigned long long)__popcll(m);\n                if (n_stg > PAS_STAGE - 64) flush(); // LDS accesses of one wavefront complete in program order\n            }\n        }\n"
     "        if (dbg) { // LM_DEBUG_PA_SEARCH: {candidates, wavefront passes, enumeration passes, anchors, clocks of the search, of the enumeration}\n"
     "            const unsigned long long nc_w = (unsigned long long)__popcll(__ballot(ci < nc)), d_2 = wall_clock64();\n"
     "            if (lane == 0) {\n"
     "                atomicAdd(dbg + 0, nc_w);\n                atomicAdd(dbg + 1, 1ull);\n                atomicAdd(dbg + 2, d_it);\n                atomicAdd(dbg + 3, d_an);\n"
     "                atomicAdd(dbg + 4, d_1 - d_0);\n                atomicAdd(dbg + 5, d_2 - d_1);\n            }\n        }\n"),
    ("    hipLaunchKernelGGL(k_pa_search, dim3(nseg8 * bps), dim3(256), 0, st, ix, tasks, wbuf, keys_cmp, vals_cmp, posoff, nvalid, cmp_tab,\n"
     "                       tab_off, tab_bits, K, min_prefix, seg_count, seg_cap, bps, cand, count, cap, outA, outB, qbits, tbits, nseg,\n"
     "                       xcd_map);\n",
     "    static const bool ps_dbg = getenv(\"LM_DEBUG_PA_SEARCH\") != nullptr;\n"
     "    static unsigned long long *d_dbg = nullptr;\n"
     "    if (ps_dbg && !d_dbg && hipMalloc((void **)&d_dbg, 8 * sizeof(unsigned long long)) != hipSuccess) d_dbg = nullptr;\n"
     "    if (ps_dbg && d_dbg) (void)hipMemsetAsync(d_dbg, 0, 8 * sizeof(unsigned long long), st);\n"
     "    hipLaunchKernelGGL(k_pa_search, dim3(nseg8 * bps), dim3(256), 0, st, ix, tasks, wbuf, keys_cmp, vals_cmp, posoff, nvalid, cmp_tab,\n"
     "                       tab_off, tab_bits, K, min_prefix, seg_count, seg_cap, bps, cand, count, cap, outA, outB, qbits, tbits, nseg,\n"
     "                       xcd_map, ps_dbg ? d_dbg : nullptr);\n"
     "    if (ps_dbg && d_dbg) {\n"
     "        unsigned long long h[8] = {0};\n"
     "        (void)hipStreamSynchronize(st);\n        (void)hipMemcpy(h, d_dbg, sizeof h, hipMemcpyDeviceToHost);\n"
     "        fprintf(stderr, \"[lm] k_pa_search: %llu candidates in %llu wavefront passes (%.1f per pass), %llu enumeration passes, %llu anchors (%.2f per enumeration pass), wavefront-ms: search %.1f, enumeration %.1f\\n\",\n"
     "                h[0], h[1], h[1] ? (double)h[0] / (double)h[1] : 0.0, h[2], h[3], h[2] ? (double)h[3] / (double)h[2] : 0.0, (double)h[4] / 1e5, (double)h[5] / 1e5);\n"
     "    }\n"),
])
print("LM_DEBUG_PA_SEARCH adopted under", root)
