#!/bin/bash
# 16-bit ring cells + predicted start widths: parity tests, then the c3-shaped workload with each on / off, and a timeline
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_wfa_mw.py tests/test_gpu_longreads.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04_tests_gpu_b.log; tail -3 gpurun_out/r04_tests_gpu_b.log
C3S="--workload c3 --genomes 20000 --queries 2000 --families 200 --steps 2 --warmup 1 --no-cpu-baseline"
timeout 600 python bench.py $C3S --tag r16_ak40 > gpurun_out/r04_c3s_p3_default.json 2> gpurun_out/r04_c3s_p3_default.err; echo "default rc=$?"
LM_WFA_R16=0 timeout 600 python bench.py $C3S --no-exclusive-step --tag r32_ak40 > gpurun_out/r04_c3s_p3_r32.json 2> gpurun_out/r04_c3s_p3_r32.err; echo "r32 rc=$?"
LM_WFA_AK_MARGIN=-1 timeout 600 python bench.py $C3S --no-exclusive-step --tag r16_noak > gpurun_out/r04_c3s_p3_noak.json 2> gpurun_out/r04_c3s_p3_noak.err; echo "noak rc=$?"
LM_WFA_AK_MARGIN=25 timeout 600 python bench.py $C3S --no-exclusive-step --tag r16_ak25 > gpurun_out/r04_c3s_p3_ak25.json 2> gpurun_out/r04_c3s_p3_ak25.err; echo "ak25 rc=$?"
LM_DEBUG=1 timeout 600 python bench.py --workload c3 --genomes 20000 --queries 2000 --families 200 --steps 1 --warmup 1 --no-cpu-baseline --no-exclusive-step --tag timeline > gpurun_out/r04_c3s_p3_timeline.json 2> gpurun_out/r04_c3s_p3_timeline.err; echo "timeline rc=$?"
grep -E "^\[lm \+" gpurun_out/r04_c3s_p3_timeline.err > gpurun_out/r04_c3s_p3_timeline.txt
python - <<'PY'
import json
for t in ("default","r32","noak","ak25"):
    try:
        d=json.loads(open("gpurun_out/r04_c3s_p3_%s.json"%t).read().strip().splitlines()[-1])
        print(t, d["value"], d["ms_per_step"], d["rows"], {k:round(v) for k,v in d["stage_ms"].items()})
        for k in d["kernels"]:
            if k["name"].startswith("k_wfa"): print("   ",k["name"],k["launches"],k["avg_ms"],k["exclusive_avg_ms"],k["exclusive_ms_per_step"])
    except Exception as e: print(t,"failed",e)
PY
