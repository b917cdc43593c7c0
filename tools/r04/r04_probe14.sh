#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -14 > gpurun_out/r04_tests_gpu_i.log; grep -E "passed|failed|FAILED" gpurun_out/r04_tests_gpu_i.log
C3S="--workload c3 --genomes 20000 --queries 2000 --families 200 --steps 3 --warmup 1 --no-cpu-baseline --no-exclusive-step"
timeout 900 python bench.py $C3S --ab-steps 3 --ab "LM_TWO_LANES=0" --tag lanes > gpurun_out/r04_c3s_p14.json 2> gpurun_out/r04_c3s_p14.err; echo "rc=$?"; grep "A/B" gpurun_out/r04_c3s_p14.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_c3s_p14.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["step_ms"], d["rows"])
PY
