#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r03_t9.log
cat gpurun_out/r03_t9.log
C3S="--workload c3 --genomes 20000 --families 200 --queries 2000 --steps 2 --warmup 1 --no-cpu-baseline"
timeout 150 python bench.py $C3S --tag fused > gpurun_out/r03_c3s_fu.json 2> gpurun_out/r03_c3s_fu.err; echo "c3s rc=$?"
echo skip
timeout 150 python bench.py --workload c2 --steps 3 --warmup 1 --no-cpu-baseline --tag fused > gpurun_out/r03_c2_fu.json 2> gpurun_out/r03_c2_fu.err; echo "c2 rc=$?"
python - <<'PY'
import json
for t in ("c3s_fu", "c3s_tp", "c2_fu"):
    try:
        p = json.loads(open("gpurun_out/r03_%s.json" % t).read().strip().split("\n")[-1])
    except Exception as e:
        print(t, "failed", e); continue
    print(t, p["value"], p["ms_per_step"], p["rows"], {k: round(v) for k, v in p["stage_ms"].items()})
    print("   lookup:", {k: p["roofline_seed_lookup"][k] for k in ("kernel", "achieved", "frac", "avg_launch_ms", "stage", "stage_ms", "stage_frac")})
    print("   ", [(k["name"], k["launches"], k["avg_ms"], k["exclusive_avg_ms"]) for k in p["kernels"] if "lookup" in k["name"]])
PY
tail -5 gpurun_out/r03_c3s_fu.err
