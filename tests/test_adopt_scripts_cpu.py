"""tools/adopt_wfa_lean2.py, tools/adopt_pa_chain_pipe.py (the staged kernels of experiments/ wired into the product sources
behind switches), tools/adopt_arena_reserve.py and tools/adopt_pa_search_debug.py apply to a COPY of the current tree: every edit is asserted against the text it replaces, so a change
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
    shutil.copy(os.path.join(ROOT, "tests", "arena_host.cpp"), dst / "tests")
    shutil.copytree(os.path.join(ROOT, "include"), dst / "include")
    for script in ("adopt_wfa_lean2.py", "adopt_pa_chain_pipe.py", "adopt_arena_reserve.py", "adopt_pa_search_debug.py"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script), "--root", str(dst)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
    k = (dst / "lexicmap_amd" / "csrc" / "lm_kernels.hip").read_text()
    assert "k_wfa_lean2<2, int16_t, false>" in k and "lm_chain2_backtrack(" in k and "pa_clear_marks_wave(" in k and "k_pa_chain_pipe" in k and "LM_DEBUG_PA_SEARCH" in k
    mw = (dst / "lexicmap_amd" / "csrc" / "lm_wfa_mw.h").read_text()
    assert "k_wfa_mw2<4, true>" in mw
    tune = (dst / "lexicmap_amd" / "csrc" / "lm_internal.h").read_text()
    for sw in ("LM_WFA_LEAN2", "LM_PA_CHAIN_PIPE", "LM_PA_PIPE_MIN", "LM_PA_CHAIN_BT_WAVE", "LM_ARENA_RESERVE_PCT"):
        assert sw in tune
    for f in ("lm_wfa_lean2.h", "lm_wfa_lean2_fwd.h", "lm_wfa_mw2.h", "lm_wfa_mw2_fwd.h", "lm_pa_chain_pipe.h", "lm_pa_chain_pipe_dp.h", "lm_pa_chain_bt.h",
              "lm_pa_chain_bt_core.h", "lm_pa_clear_tile.h"):
        assert (dst / "lexicmap_amd" / "csrc" / f).exists(), f
    assert (dst / "tests" / "test_gpu_wfa_lean2.py").exists()
    assert '"LM_WFA_LEAN2", "0"' in (dst / "tests" / "test_gpu_longreads.py").read_text()
    # and the tree they were applied to is not this one
    assert "k_wfa_lean2" not in open(os.path.join(ROOT, "lexicmap_amd", "csrc", "lm_kernels.hip")).read()
    # the reserved arena of the adopted copy over the fake device: one device allocation, then none while the slab suffices
    if os.path.isdir("/opt/rocm/include"):
        import ctypes as C
        import random
        lib_path = str(dst / "libarena_host_adopted.so")
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-DAH_HAVE_RESERVE", "-I/opt/rocm/include", "-o", lib_path,
                               str(dst / "tests" / "arena_host.cpp")])
        L = C.CDLL(lib_path)
        L.ah_arena_new.restype = C.c_void_p
        L.ah_arena_alloc.restype = C.c_void_p
        L.ah_arena_alloc.argtypes = [C.c_void_p, C.c_size_t]
        L.ah_arena_release.argtypes = [C.c_void_p, C.c_void_p]
        L.ah_arena_reserve.argtypes = [C.c_void_p, C.c_size_t]
        L.ah_arena_trim.argtypes = [C.c_void_p]
        L.ah_arena_delete.argtypes = [C.c_void_p]
        L.ah_reset.argtypes = [C.c_size_t]
        L.ah_device_mallocs.restype = C.c_long
        L.ah_device_used.restype = C.c_size_t
        MB = 1 << 20
        L.ah_reset(4096 * MB)
        a = L.ah_arena_new()
        assert L.ah_arena_reserve(a, 3000 * MB) == 1 and L.ah_device_mallocs() == 1
        rng = random.Random(3)
        for part in range(8):  # two lanes' worth of phase buffers, sizes following the data, freed in another order
            blocks = [L.ah_arena_alloc(a, rng.randint(20, 300) * MB) for _ in range(8)]
            assert all(blocks)
            rng.shuffle(blocks)
            for b in blocks:
                assert L.ah_arena_release(a, b) == 1
        assert L.ah_device_mallocs() == 1           # the device was never asked again
        assert L.ah_arena_reserve(a, 2000 * MB) == 0  # refused (the fake device is full): nothing changes
        L.ah_arena_trim(a)
        assert L.ah_device_used() == 0              # an empty reserved slab goes back like any other
        L.ah_arena_delete(a)
