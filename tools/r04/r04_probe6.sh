#!/bin/bash
# deferred tails: parity tests, then the c3-shaped workload with the switch on / off (one resident index), and a timeline
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_longreads.py tests/test_gpu_parity.py tests/test_gpu_wfa_mw.py tests/test_gpu_c4c5.py -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r04_tests_gpu_c.log; tail -5 gpurun_out/r04_tests_gpu_c.log
C3S="--workload c3 --genomes 20000 --queries 2000 --families 200 --steps 3 --warmup 1 --no-cpu-baseline"
timeout 900 python bench.py $C3S --ab-steps 3 --ab "LM_WFA_DEFER=0|LM_WFA_DEFER=1 LM_WFA_MW=0" --tag defer > gpurun_out/r04_c3s_p6.json 2> gpurun_out/r04_c3s_p6.err; echo "rc=$?"; grep "A/B" gpurun_out/r04_c3s_p6.err
LM_DEBUG=1 timeout 600 python bench.py --workload c3 --genomes 20000 --queries 2000 --families 200 --steps 1 --warmup 1 --no-cpu-baseline --no-exclusive-step --tag timeline > gpurun_out/r04_c3s_p6_timeline.json 2> gpurun_out/r04_c3s_p6_timeline.err; echo "timeline rc=$?"
grep -E "^\[lm \+" gpurun_out/r04_c3s_p6_timeline.err > gpurun_out/r04_c3s_p6_timeline.txt
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_c3s_p6.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["step_ms"], d["rows"], {k:round(v) for k,v in d["stage_ms"].items()})
for k in d["kernels"]:
    if k["name"].startswith("k_wfa"): print("   ",k["name"],k["launches"],k["avg_ms"],k["exclusive_avg_ms"],k["exclusive_ms_per_step"])
print(d["ab"])
PY
