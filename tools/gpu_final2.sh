#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for N in 2 4 8; do
  timeout 400 python bench.py --workload c3 --steps 2 --warmup 1 --shard-of $N --shard-rank 1 --no-exclusive-step > gpurun_out/r03_c3_shard_of_$N.json 2> gpurun_out/r03_c3_shard_of_$N.err; echo "shard-of $N rc=$?"
done
timeout 500 python bench.py --workload c4 --steps 2 --warmup 1 --shard-rank 0 > gpurun_out/r03_c4_shard0_of_4.json 2> gpurun_out/r03_c4_shard0_of_4.err; echo "c4 rc=$?"; tail -4 gpurun_out/r03_c4_shard0_of_4.err
timeout 500 python bench.py --workload c5 --steps 2 --warmup 1 --shard-rank 0 > gpurun_out/r03_c5_shard0_of_8.json 2> gpurun_out/r03_c5_shard0_of_8.err; echo "c5 rc=$?"; tail -4 gpurun_out/r03_c5_shard0_of_8.err
python - <<'PY'
import json
for t in ("c3_shard_of_2", "c3_shard_of_4", "c3_shard_of_8", "c4_shard0_of_4", "c5_shard0_of_8"):
    try:
        p = json.loads(open("gpurun_out/r03_%s.json" % t).read().strip().split("\n")[-1])
    except Exception as e:
        print(t, "failed", e); continue
    print(t, p["value"], p["ms_per_step"], p["rows"], p["gbp_aligned_per_s"], {k: round(v) for k, v in p["stage_ms"].items()})
    print("   shard:", p["sharding_model"])
    print("   cfg:", p["config"]["workload"], p["config"]["index_hbm_bytes"], p["config"]["seeds_resident"])
    print("   ", [(k["name"], k["launches"], k["avg_ms"]) for k in p["kernels"][:8]])
PY
