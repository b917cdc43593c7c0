#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 700 python -m pytest tests/test_gpu_two_ranks.py -m gpu -q 2>&1 | tail -5 | cut -c1-300
LM_DEBUG=1 timeout 400 python tools/loader_io_probe.py 2>&1 | grep -v "^\[lm\] \(builder\|mem\|mask\|glue\)" | tail -25 | cut -c1-300 | tee gpurun_out/r05_loader_io_probe.txt
timeout 900 python bench.py --workload c3 --genome-len 3000000 --steps 2 --warmup 2 --no-cpu-baseline --no-exclusive-step > gpurun_out/r05_c3_3mb.json 2> gpurun_out/r05_c3_3mb.err; echo "c3 3Mb rc=$?"; tail -3 gpurun_out/r05_c3_3mb.err | cut -c1-300
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r05_c3_3mb.json").read().strip().splitlines()[-1])
    print("c3 3Mb", d["value"], d["ms_per_step"], d.get("first_step_ms"), d.get("warmup_step_ms"), d["step_ms"], d["rows"], d["config"]["index_hbm_bytes"], d["config"]["workload"][:120])
except Exception as e: print("failed", e)
PY
