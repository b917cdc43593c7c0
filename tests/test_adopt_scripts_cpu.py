"""tools/adopt_wfa_lean2.py and tools/adopt_pa_chain_pipe.py (the staged kernels of experiments/ wired into the product
sources behind switches) apply to a COPY of the current tree: every edit is asserted against the text it replaces, so a change
of lexicmap_amd/csrc that the scripts do not follow fails here and not in the first minutes of a GPU session.  (That the
adopted tree BUILDS is checked by hand - a minute of hipcc - and recorded in experiments/README.md.)"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_both_adoptions_apply_to_a_copy_of_the_tree(tmp_path):
    dst = tmp_path / "tree"
    (dst / "lexicmap_amd").mkdir(parents=True)
    (dst / "tests").mkdir()
    shutil.copytree(os.path.join(ROOT, "lexicmap_amd", "csrc"), dst / "lexicmap_amd" / "csrc", ignore=shutil.ignore_patterns("*.o"))
    shutil.copy(os.path.join(ROOT, "tests", "test_gpu_longreads.py"), dst / "tests")
    for script in ("adopt_wfa_lean2.py", "adopt_pa_chain_pipe.py"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script), "--root", str(dst)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
    k = (dst / "lexicmap_amd" / "csrc" / "lm_kernels.hip").read_text()
    assert "k_wfa_lean2<2, int16_t, false>" in k and "lm_chain2_backtrack(" in k and "pa_clear_marks_wave(" in k and "k_pa_chain_pipe" in k
    mw = (dst / "lexicmap_amd" / "csrc" / "lm_wfa_mw.h").read_text()
    assert "k_wfa_mw2<4, true>" in mw
    tune = (dst / "lexicmap_amd" / "csrc" / "lm_internal.h").read_text()
    for sw in ("LM_WFA_LEAN2", "LM_PA_CHAIN_PIPE", "LM_PA_PIPE_MIN", "LM_PA_CHAIN_BT_WAVE"):
        assert sw in tune
    for f in ("lm_wfa_lean2.h", "lm_wfa_lean2_fwd.h", "lm_wfa_mw2.h", "lm_wfa_mw2_fwd.h", "lm_pa_chain_pipe.h", "lm_pa_chain_pipe_dp.h", "lm_pa_chain_bt.h",
              "lm_pa_chain_bt_core.h", "lm_pa_clear_tile.h"):
        assert (dst / "lexicmap_amd" / "csrc" / f).exists(), f
    assert (dst / "tests" / "test_gpu_wfa_lean2.py").exists()
    assert '"LM_WFA_LEAN2", "0"' in (dst / "tests" / "test_gpu_longreads.py").read_text()
    # and the tree they were applied to is not this one
    assert "k_wfa_lean2" not in open(os.path.join(ROOT, "lexicmap_amd", "csrc", "lm_kernels.hip")).read()
