#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
LM_DEBUG_ARENA=1 timeout 900 python bench.py --workload c3 --steps 9 --warmup 1 --no-cpu-baseline --no-exclusive-step --tag steady > gpurun_out/r04_c3_steady.json 2> gpurun_out/r04_c3_steady.err; echo "rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_c3_steady.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["step_ms"], d["rows"])
PY
