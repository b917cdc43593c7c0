#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r03_t3.log
cat gpurun_out/r03_t3.log
C3S="--workload c3 --genomes 20000 --families 200 --queries 2000 --steps 2 --warmup 1 --no-cpu-baseline"
timeout 600 python bench.py $C3S --tag win > gpurun_out/r03_c3s_win.json 2> gpurun_out/r03_c3s_win.err; echo "win rc=$?"
LM_NO_PIPELINE=1 LM_WFA_SERIAL=1 timeout 600 python bench.py $C3S --tag win_x > gpurun_out/r03_c3s_winx.json 2> gpurun_out/r03_c3s_winx.err; echo "winx rc=$?"
timeout 600 python bench.py --workload c2 --steps 3 --warmup 1 --no-cpu-baseline --tag win > gpurun_out/r03_c2_win.json 2> gpurun_out/r03_c2_win.err; echo "c2 rc=$?"
python - <<'PY'
import json
for t in ("c3s_win", "c3s_winx", "c2_win"):
    try:
        p = json.loads(open("gpurun_out/r03_%s.json" % t).read().strip().split("\n")[-1])
    except Exception as e:
        print(t, "failed", e); continue
    print(t, p["value"], p["ms_per_step"], p["rows"], {k: round(v) for k, v in p["stage_ms"].items()}, p["work"]["wfa_retries"])
    for k in p["kernels"][:12]: print("   ", k["name"], k["launches"], k["avg_ms"], k["ms_per_step"])
PY
tail -3 gpurun_out/r03_c3s_win.err
