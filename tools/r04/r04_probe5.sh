#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
C3S="--workload c3 --genomes 20000 --queries 2000 --families 200 --steps 4 --warmup 1 --no-cpu-baseline --no-exclusive-step"
timeout 600 python bench.py $C3S --tag grid_ak40 > gpurun_out/r04_c3s_p5_ak40.json 2> gpurun_out/r04_c3s_p5_ak40.err; echo "ak40 rc=$?"
LM_WFA_AK_MARGIN=-1 timeout 600 python bench.py $C3S --tag grid_noak > gpurun_out/r04_c3s_p5_noak.json 2> gpurun_out/r04_c3s_p5_noak.err; echo "noak rc=$?"
LM_WFA_AK_MARGIN=25 timeout 600 python bench.py $C3S --tag grid_ak25 > gpurun_out/r04_c3s_p5_ak25.json 2> gpurun_out/r04_c3s_p5_ak25.err; echo "ak25 rc=$?"
python - <<'PY'
import json
for t in ("ak40","noak","ak25"):
    try:
        d=json.loads(open("gpurun_out/r04_c3s_p5_%s.json"%t).read().strip().splitlines()[-1])
        print(t, d["value"], d["ms_per_step"], d["step_ms"], d["rows"], {k:round(v) for k,v in d["stage_ms"].items()})
        for k in d["kernels"]:
            if k["name"].startswith("k_wfa"): print("   ",k["name"],k["launches"],k["avg_ms"])
    except Exception as e: print(t,"failed",e)
PY
