"""Two ranks on ONE GPU (gloo: RCCL refuses two ranks on a device): `bench.py --gpus 2 --dist-backend gloo` spawns two processes,
each opens its genome shard (g % 2) of the same small index on the same device, searches the same batch, the rows are gathered to
rank 0 (torch.distributed: the C gather needs RCCL) and merged by lm_merge_sharded - the number of rows (and, within the rounding of the
line's rates, of aligned bases) must be those of the one-rank run.  (The N > 1 logic of the sharded search end to end; row-for-row equality of shards vs the unsharded
index: tests/test_gpu_parity.py, tests/test_gpu_c4c5.py.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(extra, more_env=None):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.pop("LM_RCCL_LIB", None)
    env.update(more_env or {})
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):  # (another test of the session may have set a rendezvous up)
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tiny", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline", "--no-exclusive-step"] + extra, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_two_ranks_on_one_gpu_give_the_rows_of_one_rank():
    one = _line([])
    two = _line(["--gpus", "2", "--dist-backend", "gloo"])
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert two["rows"] == one["rows"] and one["rows"] > 0
    a1, a2 = one["gbp_aligned_per_s"] * one["ms_per_step"], two["gbp_aligned_per_s"] * two["ms_per_step"]  # ~ aligned bases (rounded rates)
    assert abs(a1 - a2) <= 0.02 * max(a1, a2)
    assert "index-shard x2" in two["config"]["parallelism"]


def test_two_ranks_on_one_gpu_through_the_c_gather_and_the_device_merge(tmp_path):
    """the same two-rank run with the row gather and the merge behind the C-ABI (lm_comm_init / lm_gather_merge_rows), the nccl*
    symbols supplied by tests/fake_rccl.c (sockets + hipMemcpy: RCCL refuses two ranks on one device) through LM_RCCL_LIB"""
    so = str(tmp_path / "libfake_rccl.so")
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O2", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(ROOT, "tests", "fake_rccl.c"),
                           "-o", so, "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"])
    one = _line([])
    two = _line(["--gpus", "2", "--dist-backend", "gloo", "--gather", "c"], {"LM_RCCL_LIB": so, "LM_FAKE_RCCL_DIR": str(tmp_path)})
    assert two["n_gpus"] == 2 and "lm_gather_merge_rows" in two["config"]["gather"], two["config"]["gather"]
    assert two["rows"] == one["rows"] and one["rows"] > 0
    a1, a2 = one["gbp_aligned_per_s"] * one["ms_per_step"], two["gbp_aligned_per_s"] * two["ms_per_step"]
    assert abs(a1 - a2) <= 0.02 * max(a1, a2)
