#!/bin/bash
# the default bench line (what the driver runs): python bench.py  (C3, 3 steps after 3 warm-up steps), with the committed counter passes in place
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python bench.py > gpurun_out/r04_c3_bench.json 2> gpurun_out/r04_c3_bench.err; echo "rc=$?"; tail -2 gpurun_out/r04_c3_bench.err | cut -c1-200
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_c3_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["step_ms"], d["rows"], d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"]["full_index_rows_equal"])
PY
