#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r03_t7.log
cat gpurun_out/r03_t7.log
timeout 300 python bench.py --workload tiny --steps 2 --warmup 1 > gpurun_out/r03_tiny.json 2> gpurun_out/r03_tiny.err; echo "tiny rc=$?"; tail -3 gpurun_out/r03_tiny.err
timeout 600 python bench.py --workload c3 --genomes 8000 --families 80 --queries 400 --steps 2 --warmup 1 --shard-of 4 --shard-rank 1 > gpurun_out/r03_so4.json 2> gpurun_out/r03_so4.err; echo "so4 rc=$?"; tail -3 gpurun_out/r03_so4.err
timeout 600 python bench.py --workload c4 --genomes 8000 --families 81 --queries 8 --steps 2 --warmup 1 > gpurun_out/r03_c4mini.json 2> gpurun_out/r03_c4mini.err; echo "c4mini rc=$?"; tail -3 gpurun_out/r03_c4mini.err
timeout 600 python bench.py --workload c5 --genomes 16000 --families 161 --queries 400 --steps 2 --warmup 1 > gpurun_out/r03_c5mini.json 2> gpurun_out/r03_c5mini.err; echo "c5mini rc=$?"; tail -3 gpurun_out/r03_c5mini.err
python - <<'PY'
import json
for t in ("tiny", "so4", "c4mini", "c5mini"):
    try:
        p = json.loads(open("gpurun_out/r03_%s.json" % t).read().strip().split("\n")[-1])
    except Exception as e:
        print(t, "failed", e); continue
    print(t, p["value"], p["ms_per_step"], p["rows"], p["config"]["workload"][:80])
    print("   roofline:", {k: p["roofline"][k] for k in ("kernel", "achieved", "frac", "avg_launch_ms", "duration_kind", "avg_launch_ms_coscheduled")})
    print("   pipeline:", p["roofline_pipeline"])
    print("   shard:", p["sharding_model"])
    print("   ", [(k["name"], k["launches"], k["avg_ms"], k["exclusive_avg_ms"]) for k in p["kernels"][:8]])
PY
