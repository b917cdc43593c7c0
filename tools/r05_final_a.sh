#!/bin/bash
# round-5 final batch, part A (one gpurun call): the GPU parity suite, then the C3 profile set - kernel trace, FETCH / WRITE / SQ
# counter passes (separate runs, never together with tracing), then the bench line that reads them (steps 3, warmup 3 = the
# driver's own command)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/r05_tests_gpu.log; grep -E "passed|failed" gpurun_out/r05_tests_gpu.log
bash tools/profile.sh c3 2 2>&1 | grep -E "^workload|rank 0|^k_wfa_lean |^k_pa_|^k_lookup" | cut -c1-300 | head -16
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r05_c3_bench.json").read().strip().splitlines()[-1])
    print("c3", d["value"], d["ms_per_step"], "first", d.get("first_step_ms"), d.get("warmup_step_ms"), d.get("step_ms"), d["rows"])
    print(json.dumps(d["roofline"])[:1200])
    print(json.dumps(d.get("cpu_baseline"))[:1500])
except Exception as e: print("c3 failed", e)
PY
