#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline --no-exclusive-step --ab-steps 2 --ab "LM_TWO_LANES=1|LM_WFA_DEFER=1|LM_TWO_LANES=1 LM_WFA_DEFER=1" > gpurun_out/r04_c3_ab2.json 2> gpurun_out/r04_c3_ab2.err; echo "rc=$?"
grep -E "A/B|index ready" gpurun_out/r04_c3_ab2.err | cut -c1-160
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_c3_ab2.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["step_ms"], d["rows"], {k:round(v) for k,v in d["stage_ms"].items()})
print(d["ab"])
PY
