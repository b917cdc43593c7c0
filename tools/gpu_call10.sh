#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r03_t10.log
cat gpurun_out/r03_t10.log
df -h /tmp | tail -1
timeout 400 python bench.py --workload c2 --genomes 2500 --families 25 --queries 500 --steps 2 --warmup 1 --no-cpu-baseline --loader-check > gpurun_out/r03_loader.json 2> gpurun_out/r03_loader.err; echo "loader rc=$?"; grep "loader check\|Error\|error" gpurun_out/r03_loader.err | tail -3
C3S="--workload c3 --genomes 20000 --families 200 --queries 2000 --steps 2 --warmup 1 --no-cpu-baseline"
timeout 200 python bench.py $C3S --tag kmers > gpurun_out/r03_c3s_km.json 2> gpurun_out/r03_c3s_km.err; echo "c3s rc=$?"
python - <<'PY'
import json
for t in ("loader", "c3s_km"):
    try:
        p = json.loads(open("gpurun_out/r03_%s.json" % t).read().strip().split("\n")[-1])
    except Exception as e:
        print(t, "failed", e); continue
    print(t, p["value"], p["ms_per_step"], p["rows"], {k: round(v) for k, v in p["stage_ms"].items()})
    print("   loader:", p.get("loader"))
    print("   ", [(k["name"], k["launches"], k["avg_ms"], k["exclusive_avg_ms"]) for k in p["kernels"] if k["name"] in ("k_extract_kmers", "k_chain1", "k_mask")])
PY
