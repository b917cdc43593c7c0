#!/bin/bash
# the round-end batch on the GPU box (one gpurun call), most important first: parity tests, the C3 profile set (kernel trace,
# FETCH / WRITE / SQ passes, then the bench line that reads them), the C2 profile set, one shard of the 8-GPU C3 run, the C4 and
# C5 shards, shards of the 2- and 4-GPU C3 runs, the loader at C2 size
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/r04_tests_gpu.log
grep -E "passed|failed" gpurun_out/r04_tests_gpu.log
bash tools/profile.sh c3 3 2>&1 | grep -E "^workload|rank 0|^k_wfa_lean |^k_pa_search|^k_lookup" | cut -c1-300 | head -14
bash tools/profile.sh c2 3 2>&1 | grep -E "^workload|rank 0" | cut -c1-300 | head -4
timeout 300 python bench.py --workload c3 --steps 2 --warmup 1 --shard-of 8 --shard-rank 1 --no-exclusive-step > gpurun_out/r04_c3_shard_of_8.json 2> gpurun_out/r04_c3_shard_of_8.err; echo "shard-of 8 rc=$?"
timeout 400 python bench.py --workload c4 --steps 2 --warmup 1 --shard-rank 0 > gpurun_out/r04_c4_shard0_of_4.json 2> gpurun_out/r04_c4_shard0_of_4.err; echo "c4 rc=$?"
timeout 400 python bench.py --workload c5 --steps 2 --warmup 1 --shard-rank 0 > gpurun_out/r04_c5_shard0_of_8.json 2> gpurun_out/r04_c5_shard0_of_8.err; echo "c5 rc=$?"
timeout 300 python bench.py --workload c3 --steps 2 --warmup 1 --shard-of 4 --shard-rank 1 --no-exclusive-step > gpurun_out/r04_c3_shard_of_4.json 2> gpurun_out/r04_c3_shard_of_4.err; echo "shard-of 4 rc=$?"
timeout 400 python bench.py --workload c3 --steps 2 --warmup 1 --shard-of 2 --shard-rank 1 --no-exclusive-step > gpurun_out/r04_c3_shard_of_2.json 2> gpurun_out/r04_c3_shard_of_2.err; echo "shard-of 2 rc=$?"
timeout 500 python bench.py --workload c2 --steps 1 --warmup 1 --no-cpu-baseline --no-exclusive-step --loader-check > gpurun_out/r04_c2_loader.json 2> gpurun_out/r04_c2_loader.err; echo "loader rc=$?"; grep "loader check" gpurun_out/r04_c2_loader.err | cut -c1-400
python - <<'PY'
import json
for f in ("r04_c3_bench","r04_c2_bench","r04_c3_shard_of_8","r04_c4_shard0_of_4","r04_c5_shard0_of_8","r04_c3_shard_of_4","r04_c3_shard_of_2"):
    try:
        d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d.get("step_ms"), d["rows"], (d.get("sharding_model") or {}).get("predicted_queries_per_s"))
    except Exception as e: print(f, "failed", e)
PY
