#!/bin/bash
# round 6, final evidence on frozen sources (one gpurun call): GPU suite, C3 and C2 passes (trace + three counter passes over a
# warm step + bench line), the shard models, C4 / C5 shards, the loader check, and the C3 line as the driver runs it (more steps)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd $(dirname $0)/.. && pwd)
cd $R; mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r06_tests_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/r06_tests_gpu.log | tail -1
bash tools/profile.sh c3 2 > gpurun_out/r06_profile_c3.log 2>&1
bash tools/profile.sh c2 3 > gpurun_out/r06_profile_c2.log 2>&1
for n in 2 4 8; do python bench.py --shard-of $n --steps 5 --warmup 5 > gpurun_out/r06_c3_shard_of_$n.json 2> gpurun_out/r06_c3_shard_of_$n.err; done
python bench.py --workload c4 --steps 3 --warmup 3 > gpurun_out/r06_c4_shard0_of_4.json 2> gpurun_out/r06_c4_shard0_of_4.err
python bench.py --workload c5 --steps 3 --warmup 3 > gpurun_out/r06_c5_shard0_of_8.json 2> gpurun_out/r06_c5_shard0_of_8.err
python bench.py --workload c2 --loader-check --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/r06_c2_loader.json 2> gpurun_out/r06_c2_loader.err
python bench.py --steps 10 --warmup 5 > gpurun_out/r06_c3_bench_10steps.json 2> gpurun_out/r06_c3_bench_10steps.err
python - <<PY
import json
for f in ("c3_bench","c3_bench_10steps","c2_bench","c3_shard_of_2","c3_shard_of_4","c3_shard_of_8","c4_shard0_of_4","c5_shard0_of_8","c2_loader"):
    try:
        d=json.load(open("gpurun_out/r06_%s.json"%f)); m=d.get("sharding_model") or {}
        print(f, d["value"], d["ms_per_step"], d.get("step_ms"), (d.get("roofline") or {}).get("traffic"), {k:m.get(k) for k in ("merge_ms","predicted_step_ms","predicted_queries_per_s")}, (d.get("loader") or {}).get("open_s"))
    except Exception as e: print(f, "failed", e)
PY
