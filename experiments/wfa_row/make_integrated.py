#!/usr/bin/env python3
"""experiments/wfa_row/make_integrated.py - the product sources with k_wfa_mw wired into the 512 / 1024-diagonal WFA passes, in a
scratch copy (experiments/csrc_mw, library experiments/lib_mw/liblexicmap_hip.so) so that lexicmap_amd/csrc - whose
hash the committed counter passes are tied to - stays as measured.  Writes integrate_mw.patch (the diff to apply next round: `git apply experiments/wfa_row/integrate_mw.patch`; then add lm_wfa_mw.h lm_wfa_mw_fwd.h to the Makefile's lm_kernels.o rule).
Run the GPU tests against the scratch library with LEXICMAP_HIP_LIB=experiments/lib_mw/liblexicmap_hip.so."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SRC = os.path.join(ROOT, "lexicmap_amd", "csrc")
DST = os.path.join(ROOT, "experiments", "csrc_mw")  # same depth as lexicmap_amd/csrc: the relative includes hold


def sub(path, old, new, count=1):
    s = open(path).read()
    assert s.count(old) >= 1, (path, old[:60])
    open(path, "w").write(s.replace(old, new, count))


def main():
    shutil.rmtree(DST, ignore_errors=True)
    os.makedirs(DST)
    for f in os.listdir(SRC):
        if f.endswith((".hip", ".h", ".cpp")) or f == "Makefile":
            shutil.copy(os.path.join(SRC, f), os.path.join(DST, f))
    shutil.copy(os.path.join(HERE, "wfa_mw_fwd.h"), os.path.join(DST, "lm_wfa_mw_fwd.h"))
    shutil.copy(os.path.join(HERE, "lm_wfa_mw.h"), os.path.join(DST, "lm_wfa_mw.h"))
    k = os.path.join(DST, "lm_kernels.hip")
    sub(k, "// ------------------------------------------------------------------------------------------------------------\n// host-callable launchers\n",
        '#include "lm_wfa_mw.h"\n\n// ------------------------------------------------------------------------------------------------------------\n// host-callable launchers\n')
    h = os.path.join(DST, "lm_kernels.h")
    sub(h, "// wavefronts wider than the LDS ring (status 3 from launch_wfa)",
        "// k_wfa_mw<nc / 4, win>: the same passes for nc = 8 / 16 by a workgroup of four wavefronts per alignment\n"
        "int wfa_mw_resident_blocks(int device, int seq_words, int nc, bool win);\n"
        "void launch_wfa_mw(hipStream_t st, const WfaIn *in, int64_t n, const int32_t *todo, int64_t ntodo, int nblocks, int32_t *hdr_pool,\n"
        "                   int64_t hdr_stride, uint8_t *arena_pool, int64_t arena_stride, uint64_t *ops_pool, unsigned int *queue, int seq_words,\n"
        "                   int want_ops, WfaOut *out, int nc, bool win);\n\n"
        "// wavefronts wider than the LDS ring (status 3 from launch_wfa)")
    t = os.path.join(DST, "lm_internal.h")
    sub(t, "    int wfa_serial = 0;      // LM_WFA_SERIAL=1",
        "    int wfa_mw = 1;          // 512 / 1024-diagonal passes by a workgroup of four wavefronts per alignment (LM_WFA_MW=0: one wavefront)\n"
        "    int wfa_serial = 0;      // LM_WFA_SERIAL=1")
    sub(t, "        no_pipeline = getenv(\"LM_NO_PIPELINE\") != nullptr;\n",
        "        no_pipeline = getenv(\"LM_NO_PIPELINE\") != nullptr;\n        if (const char *e = getenv(\"LM_WFA_MW\")) wfa_mw = atoi(e) != 0;\n")
    p = os.path.join(DST, "lm_pipeline.hip")
    sub(p, "        const int resident = wfa_resident_blocks(ix->device, seq_words, nc, use_win);\n"
           "        int nblocks = (int)std::min<int64_t>(m, std::max<int64_t>(256, (int64_t)resident * ix->tune.wfa_resident_pct / 100));\n",
           "        // the 512 / 1024-diagonal passes are a handful of long alignments the round waits for: a workgroup of four wavefronts each\n"
           "        const bool mw = ix->tune.wfa_mw && nc >= 8;\n"
           "        const int resident = mw ? wfa_mw_resident_blocks(ix->device, seq_words, nc, use_win) : wfa_resident_blocks(ix->device, seq_words, nc, use_win);\n"
           "        int nblocks = (int)std::min<int64_t>(m, std::max<int64_t>(mw ? 1 : 256, (int64_t)resident * ix->tune.wfa_resident_pct / 100));\n")
    sub(p, "            static const char *const names[2][5] = {{\"k_wfa_lean64\", \"k_wfa_lean\", \"k_wfa_lean256\", \"k_wfa_lean512\", \"k_wfa_lean1024\"},\n",
           "            static const char *const names[4][5] = {{\"k_wfa_lean64\", \"k_wfa_lean\", \"k_wfa_lean256\", \"k_wfa_lean512\", \"k_wfa_lean1024\"},\n"
           "                                                    {\"k_wfa_win64\", \"k_wfa_win128\", \"k_wfa_win256\", \"k_wfa_win512\", \"k_wfa_win1024\"},\n"
           "                                                    {\"\", \"\", \"\", \"k_wfa_mw512\", \"k_wfa_mw1024\"},\n"
           "                                                    {\"\", \"\", \"\", \"k_wfa_mww512\", \"k_wfa_mww1024\"}};\n"
           "            static const char *const unused_names[1][5] = {\n")
    sub(p, "            Prof p(ix, names[use_win ? 1 : 0][nc == 16 ? 4 : nc == 8 ? 3 : nc == 4 ? 2 : nc == 1 ? 0 : 1], wfa_bytes(in, items));\n"
           "            launch_wfa(S(ix), a.wfa_in.p, n, lc.todo.p, m, nblocks, lc.hdr_pool.p, entries * 2, (uint8_t *)lc.arena_pool.p, bytes,\n"
           "                       a.ops_pool.p, lc.queue.p, seq_words, want_ops ? 1 : 0, a.wfa_out.p, nc, use_win);\n",
           "            (void)unused_names;\n"
           "            Prof p(ix, names[mw ? (use_win ? 3 : 2) : use_win ? 1 : 0][nc == 16 ? 4 : nc == 8 ? 3 : nc == 4 ? 2 : nc == 1 ? 0 : 1], wfa_bytes(in, items));\n"
           "            if (mw)\n"
           "                launch_wfa_mw(S(ix), a.wfa_in.p, n, lc.todo.p, m, nblocks, lc.hdr_pool.p, entries * 2, (uint8_t *)lc.arena_pool.p, bytes,\n"
           "                              a.ops_pool.p, lc.queue.p, seq_words, want_ops ? 1 : 0, a.wfa_out.p, nc, use_win);\n"
           "            else\n"
           "                launch_wfa(S(ix), a.wfa_in.p, n, lc.todo.p, m, nblocks, lc.hdr_pool.p, entries * 2, (uint8_t *)lc.arena_pool.p, bytes,\n"
           "                           a.ops_pool.p, lc.queue.p, seq_words, want_ops ? 1 : 0, a.wfa_out.p, nc, use_win);\n")
    m = os.path.join(DST, "Makefile")
    sub(m, "OUT = ../liblexicmap_hip.so", "OUT = ../lib_mw/liblexicmap_hip.so")  # same file name: tests that link with -llexicmap_hip work on it
    sub(m, "lm_kernels.o: lm_kernels.hip lm_kernels.h lm_algos.h", "lm_kernels.o: lm_kernels.hip lm_kernels.h lm_algos.h lm_wfa_mw.h lm_wfa_mw_fwd.h")
    # one unified diff with a/ b/ labels on the product paths: `git apply experiments/wfa_row/integrate_mw.patch` from the root
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
    open(os.path.join(HERE, "integrate_mw.patch"), "w").write(patch)
    print("patch: %d lines" % patch.count("\n"))
    if "--no-build" not in sys.argv:
        os.makedirs(os.path.join(ROOT, "experiments", "lib_mw"), exist_ok=True)
        subprocess.check_call(["make", "-s", "-C", DST, "-j4"])
        print("built", os.path.join(ROOT, "experiments", "lib_mw", "liblexicmap_hip.so"))


if __name__ == "__main__":
    main()
