#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r03_t8.log
cat gpurun_out/r03_t8.log
timeout 600 python bench.py --workload c4 --genomes 8000 --families 81 --queries 8 --steps 2 --warmup 1 > gpurun_out/r03_c4mini.json 2> gpurun_out/r03_c4mini.err; echo "c4mini rc=$?"; tail -3 gpurun_out/r03_c4mini.err
C3S="--workload c3 --genomes 20000 --families 200 --queries 2000 --steps 2 --warmup 1 --no-cpu-baseline"
timeout 600 python bench.py $C3S --tag chainwave > gpurun_out/r03_c3s_cw.json 2> gpurun_out/r03_c3s_cw.err; echo "c3s rc=$?"
LM_CHAIN1_LANES=1 timeout 600 python bench.py $C3S --tag chainlanes > gpurun_out/r03_c3s_cl.json 2> gpurun_out/r03_c3s_cl.err; echo "c3s rc=$?"
python - <<'PY'
import json
for t in ("c4mini", "c3s_cw", "c3s_cl"):
    try:
        p = json.loads(open("gpurun_out/r03_%s.json" % t).read().strip().split("\n")[-1])
    except Exception as e:
        print(t, "failed", e); continue
    print(t, p["value"], p["ms_per_step"], p["rows"], {k: round(v) for k, v in p["stage_ms"].items()})
    print("   ", [(k["name"], k["launches"], k["avg_ms"], k["exclusive_avg_ms"]) for k in p["kernels"][:12]])
PY
