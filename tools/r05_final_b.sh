#!/bin/bash
# round-5 final batch, part B: the C2 profile set, one shard each of the 8- / 4- / 2-GPU C3 runs in steady state, the C4 and C5
# shards, the loader at C2 size
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/profile.sh c2 3 2>&1 | grep -E "^workload|rank 0" | cut -c1-300 | head -4
timeout 300 python bench.py --workload c3 --steps 3 --warmup 3 --shard-of 8 --shard-rank 1 --no-exclusive-step > gpurun_out/r05_c3_shard_of_8.json 2> gpurun_out/r05_c3_shard_of_8.err; echo "shard-of 8 rc=$?"
timeout 400 python bench.py --workload c3 --steps 3 --warmup 3 --shard-of 4 --shard-rank 1 --no-exclusive-step > gpurun_out/r05_c3_shard_of_4.json 2> gpurun_out/r05_c3_shard_of_4.err; echo "shard-of 4 rc=$?"
timeout 500 python bench.py --workload c3 --steps 3 --warmup 3 --shard-of 2 --shard-rank 1 --no-exclusive-step > gpurun_out/r05_c3_shard_of_2.json 2> gpurun_out/r05_c3_shard_of_2.err; echo "shard-of 2 rc=$?"
timeout 400 python bench.py --workload c4 --steps 3 --warmup 3 --shard-rank 0 > gpurun_out/r05_c4_shard0_of_4.json 2> gpurun_out/r05_c4_shard0_of_4.err; echo "c4 rc=$?"
timeout 400 python bench.py --workload c5 --steps 3 --warmup 3 --shard-rank 0 > gpurun_out/r05_c5_shard0_of_8.json 2> gpurun_out/r05_c5_shard0_of_8.err; echo "c5 rc=$?"
LM_DEBUG=1 timeout 500 python bench.py --workload c2 --steps 1 --warmup 1 --no-cpu-baseline --no-exclusive-step --loader-check > gpurun_out/r05_c2_loader.json 2> gpurun_out/r05_c2_loader.err; echo "loader rc=$?"; grep -E "loader check|loader:" gpurun_out/r05_c2_loader.err | cut -c1-400 | tail -6
python - <<'PY'
import json
for f in ("r05_c2_bench","r05_c3_shard_of_8","r05_c3_shard_of_4","r05_c3_shard_of_2","r05_c4_shard0_of_4","r05_c5_shard0_of_8","r05_c2_loader"):
    try:
        d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], "first", d.get("first_step_ms"), d.get("step_ms"), d["rows"], (d.get("sharding_model") or {}).get("predicted_queries_per_s"), (d.get("loader") or {}).get("open_s"))
    except Exception as e: print(f, "failed", e)
PY
