#!/usr/bin/env python3
"""adopt_wfa_lean2.py [--root DIR]: puts the staged k_wfa_lean2 (experiments/wfa_lean2: the restructured forward pass of the
single-wavefront WFA kernel, and k_wfa_mw2: the same for the workgroup kernel) into the product sources under DIR (default: this repository): the two headers into
lexicmap_amd/csrc, the kernel family selectable per handle (LM_WFA_LEAN2, default on; 0 = k_wfa_lean), the dynamic LDS of the
whole-sequence form, the Makefile dependencies.  Every edit is asserted against the text it replaces.  Round 5: run it, build,
run experiments/wfa_lean2/gpu_check.py on the GPU, then the GPU test suite and the bench lines.  (Checked in round 4 on a copy of
the tree: the library builds.)"""
import os
import shutil
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) >= 3 and sys.argv[1] == "--root":
    root = os.path.abspath(sys.argv[2])
src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "experiments", "wfa_lean2")
csrc = os.path.join(root, "lexicmap_amd", "csrc")


def edit(path, pairs):
    s = open(path).read()
    for old, new in pairs:
        assert s.count(old) == 1, (path, old[:60], s.count(old))
        s = s.replace(old, new)
    open(path, "w").write(s)


for f in ("wfa_lean2_fwd.h", "lm_wfa_lean2.h", "wfa_mw2_fwd.h", "lm_wfa_mw2.h"):
    t = open(os.path.join(src, f)).read().replace("experiments/wfa_lean2/" + f, f).replace("(STAGED for round 5)", "").replace(
        "STAGED for round 5: ", "")
    open(os.path.join(csrc, "lm_" + f if not f.startswith("lm_") else f), "w").write(t)
edit(os.path.join(csrc, "lm_wfa_lean2.h"), [('#include "wfa_lean2_fwd.h"', '#include "lm_wfa_lean2_fwd.h"')])
edit(os.path.join(csrc, "lm_wfa_mw2.h"), [('#include "wfa_mw2_fwd.h"', '#include "lm_wfa_mw2_fwd.h"')])
edit(os.path.join(csrc, "lm_wfa_mw2_fwd.h"), [('#include "wfa_lean2_fwd.h"', '#include "lm_wfa_lean2_fwd.h"')])
# the workgroup kernels: both families are included by lm_wfa_mw.h (its WR_* macros are theirs), selected by the same switch
edit(os.path.join(csrc, "lm_wfa_mw.h"), [
    ('#include "lm_wfa_mw_fwd.h"\n', '#include "lm_wfa_mw_fwd.h"\n#include "lm_wfa_lean2.h"\n#include "lm_wfa_mw2.h"\n'),
    ("static WfaMwFn wfa_mw_fn(int ncw, bool win) {\n",
     "static WfaMwFn wfa_mw_fn(int ncw, bool win, bool lean2 = false) {\n"
     "    if (lean2) { // the restructured forward pass (lm_wfa_mw2.h)\n"
     "        if (win) return ncw == 4 ? k_wfa_mw2<4, true> : k_wfa_mw2<2, true>;\n"
     "        return ncw == 4 ? k_wfa_mw2<4, false> : k_wfa_mw2<2, false>;\n"
     "    }\n"),
    ("static size_t wfa_mw_dyn_lds(int seq_words, bool win) { return win ? 0 : (size_t)(2 * (seq_words + 2)) * sizeof(uint32_t); }",
     "static size_t wfa_mw_dyn_lds(int seq_words, bool win) { return win ? 0 : (size_t)(2 * (seq_words + 2) + 1) * sizeof(uint32_t); } // (k_wfa_mw2: one word in front)"),
    ("int wfa_mw_resident_blocks(int device, int seq_words, int nc, bool win) {", "int wfa_mw_resident_blocks(int device, int seq_words, int nc, bool win, bool lean2) {"),
    ("(const void *)wfa_mw_fn(nc / 4, win), MW_THREADS,", "(const void *)wfa_mw_fn(nc / 4, win, lean2), MW_THREADS,"),
    ("                   int want_ops, WfaOut *out, int nc, bool win) {\n    hipLaunchKernelGGL(wfa_mw_fn(nc / 4, win),",
     "                   int want_ops, WfaOut *out, int nc, bool win, bool lean2) {\n    hipLaunchKernelGGL(wfa_mw_fn(nc / 4, win, lean2),"),
])

edit(os.path.join(csrc, "lm_kernels.hip"), [
    ("static WfaLeanFn wfa_lean_fn(int nc, bool win, bool r16) {\n",
     "static WfaLeanFn wfa_lean2_fn(int nc, bool win, bool r16) { // the restructured forward pass (lm_wfa_lean2.h)\n"
     "    if (r16 && !win && nc == 2) return k_wfa_lean2<2, int16_t, false>;\n"
     "    if (r16 && !win && nc == 4) return k_wfa_lean2<4, int16_t, false>;\n"
     "    switch (nc) {\n"
     "    case 16: return win ? k_wfa_lean2<16, int32_t, true> : k_wfa_lean2<16, int32_t, false>;\n"
     "    case 8: return win ? k_wfa_lean2<8, int32_t, true> : k_wfa_lean2<8, int32_t, false>;\n"
     "    case 4: return win ? k_wfa_lean2<4, int32_t, true> : k_wfa_lean2<4, int32_t, false>;\n"
     "    case 1: return win ? k_wfa_lean2<1, int32_t, true> : k_wfa_lean2<1, int32_t, false>;\n"
     "    default: return win ? k_wfa_lean2<2, int32_t, true> : k_wfa_lean2<2, int32_t, false>;\n"
     "    }\n"
     "}\n"
     "static WfaLeanFn wfa_lean_fn(int nc, bool win, bool r16, bool lean2 = false) {\n"
     "    if (lean2) return wfa_lean2_fn(nc, win, r16);\n"),
    ("    return win ? 0 : (size_t)(2 * (seq_words + 1) + 2) * sizeof(uint32_t); // extension may read one word past)\n",
     "    return win ? 0 : (size_t)(2 * (seq_words + 2) + 1) * sizeof(uint32_t); // extension may read one word past; k_wfa_lean2: one word in front)\n"),
    ("int wfa_resident_blocks(int device, int seq_words, int nc, bool win, bool r16) {\n",
     "int wfa_resident_blocks(int device, int seq_words, int nc, bool win, bool r16, bool lean2) {\n"),
    ("(const void *)wfa_lean_fn(nc, win, r16), 64,", "(const void *)wfa_lean_fn(nc, win, r16, lean2), 64,"),
    ("WfaOut *out, int nc, bool win, bool r16, unsigned long long *dbg) {\n    hipLaunchKernelGGL(wfa_lean_fn(nc, win, r16),",
     "WfaOut *out, int nc, bool win, bool r16, unsigned long long *dbg, bool lean2) {\n    hipLaunchKernelGGL(wfa_lean_fn(nc, win, r16, lean2),"),
])
edit(os.path.join(csrc, "lm_kernels.h"), [
    ("int wfa_mw_resident_blocks(int device, int seq_words, int nc, bool win);", "int wfa_mw_resident_blocks(int device, int seq_words, int nc, bool win, bool lean2 = false);"),
    ("                   int want_ops, WfaOut *out, int nc, bool win);", "                   int want_ops, WfaOut *out, int nc, bool win, bool lean2 = false);"),
    ("int wfa_resident_blocks(int device, int seq_words, int nc, bool win, bool r16 = false);",
     "int wfa_resident_blocks(int device, int seq_words, int nc, bool win, bool r16 = false, bool lean2 = false);"),
    ("                unsigned long long *dbg = nullptr); // dbg: 6 words per workgroup (LM_DEBUG_WFA_WAVES)",
     "                unsigned long long *dbg = nullptr, bool lean2 = false); // dbg: 6 words per workgroup (LM_DEBUG_WFA_WAVES); lean2: k_wfa_lean2"),
])
edit(os.path.join(csrc, "lm_internal.h"), [
    ("    int wfa_mw = 1;          // 512 / 1024-diagonal passes",
     "    int wfa_lean2 = 1;       // the single-wavefront WFA passes by k_wfa_lean2 (restructured forward pass); LM_WFA_LEAN2=0: k_wfa_lean\n"
     "    int wfa_mw = 1;          // 512 / 1024-diagonal passes"),
    ('        if (const char *e = getenv("LM_WFA_R16")) wfa_r16 = atoi(e) != 0;\n',
     '        if (const char *e = getenv("LM_WFA_R16")) wfa_r16 = atoi(e) != 0;\n        if (const char *e = getenv("LM_WFA_LEAN2")) wfa_lean2 = atoi(e) != 0;\n'),
])
edit(os.path.join(csrc, "lm_pipeline.hip"), [
    ("ix->tune.wfa_r16 && wfa_r16_ok(cw[c], first_nc[c], win[c]))) * per * 9 / 8;",
     "ix->tune.wfa_r16 && wfa_r16_ok(cw[c], first_nc[c], win[c]), ix->tune.wfa_lean2 != 0)) * per * 9 / 8;"),
    (": wfa_resident_blocks(ix->device, seq_words, nc, use_win, r16);", ": wfa_resident_blocks(ix->device, seq_words, nc, use_win, r16, ix->tune.wfa_lean2 != 0);"),
    ("mw ? wfa_mw_resident_blocks(ix->device, seq_words, nc, use_win) :", "mw ? wfa_mw_resident_blocks(ix->device, seq_words, nc, use_win, ix->tune.wfa_lean2 != 0) :"),
    ("a.ops_pool.p, lc.queue.p, seq_words, want_ops ? 1 : 0, a.wfa_out.p, nc, use_win);", "a.ops_pool.p, lc.queue.p, seq_words, want_ops ? 1 : 0, a.wfa_out.p, nc, use_win, ix->tune.wfa_lean2 != 0);"),
    ("nc, use_win, r16, wave_dbg ? lc.dbg.p : nullptr);", "nc, use_win, r16, wave_dbg ? lc.dbg.p : nullptr, ix->tune.wfa_lean2 != 0);"),
])
edit(os.path.join(csrc, "Makefile"), [
    ("lm_kernels.o: lm_kernels.hip lm_kernels.h lm_algos.h lm_wfa_mw.h lm_wfa_mw_fwd.h",
     "lm_kernels.o: lm_kernels.hip lm_kernels.h lm_algos.h lm_wfa_mw.h lm_wfa_mw_fwd.h lm_wfa_lean2.h lm_wfa_lean2_fwd.h lm_wfa_mw2.h lm_wfa_mw2_fwd.h"),
])
# the forced-path GPU test (every instantiation of both families against the oracle, LM_WFA_LEAN2 on and off)
open(os.path.join(root, "tests", "test_gpu_wfa_lean2.py"), "w").write('''"""GPU parity of k_wfa_lean2 / k_wfa_mw2 (the restructured forward pass of the WFA kernels, switch LM_WFA_LEAN2): every
instantiation forced through lm_wfa_batch against the oracle's lmo_wfa_align - ring widths 64-1024 diagonals, 16- and 32-bit
cells, whole sequences and sliding windows, drifting wavefronts (the ring is recentred), pairs that outgrow a ring - and the
same with the switch off (k_wfa_lean / k_wfa_mw).  The pairs and the checks: experiments/wfa_lean2/gpu_check.py."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def test_every_instantiation_equals_the_oracle_with_the_switch_on_and_off():
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("wfa_lean2_gpu_check", os.path.join(here, "..", "experiments", "wfa_lean2", "gpu_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad, report = mod.main(timing=False)
    assert bad == 0, report
    on = [v["kernels"] for k, v in report.items() if '"LM_WFA_LEAN2": "1"' in k]
    assert on and all(v for v in on)
''')
# the rows of the long-read fixture with the new switches off (tests/test_gpu_longreads.py: every pair of device paths agrees)
t = os.path.join(root, "tests", "test_gpu_longreads.py")
if os.path.exists(t):
    edit(t, [('for var, off in (("LM_WFA_MW", "0"), ', 'for var, off in (("LM_WFA_MW", "0"), (\"LM_WFA_LEAN2\", \"0\"), ')])
t = os.path.join(root, "tests", "test_adopt_scripts_cpu.py")  # (checks that the scripts apply to the UNadopted tree: done with)
if os.path.exists(t):
    os.remove(t)
print("k_wfa_lean2 adopted under", root)
