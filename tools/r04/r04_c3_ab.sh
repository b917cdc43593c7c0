#!/bin/bash
# C3 (the headline workload) with this round's kernels, and the same resident index under the switches that turn each of them off
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python bench.py --workload c3 --steps 3 --warmup 1 --ab-steps 2 --ab "LM_WFA_AK_MARGIN=-1|LM_WFA_AK_MARGIN=-1 LM_WFA_R16=0|LM_WFA_AK_MARGIN=-1 LM_WFA_R16=0 LM_WFA_MW=0|LM_WFA_AK_MARGIN=-1 LM_PA_CHAIN_RING=0" > gpurun_out/r04_c3_ab.json 2> gpurun_out/r04_c3_ab.err; echo "rc=$?"
grep -E "A/B|full-index|step" gpurun_out/r04_c3_ab.err | tail -12
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_c3_ab.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["step_ms"], d["rows"], {k:round(v) for k,v in d["stage_ms"].items()})
for k in d["kernels"][:14]: print("   ",k["name"],k["launches"],k["avg_ms"],k["exclusive_avg_ms"],k["exclusive_ms_per_step"])
print(d["ab"]); print({k:v for k,v in d["cpu_baseline"].items() if k!="sample"})
PY
