#!/bin/bash
# full GPU suite on the current tree, then one C4 shard (LDS-ring chaining DP on / off on the same resident index)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r04_tests_gpu_e.log; tail -4 gpurun_out/r04_tests_gpu_e.log
timeout 900 python bench.py --workload c4 --steps 2 --warmup 1 --shard-rank 0 --ab-steps 2 --ab "LM_PA_CHAIN_RING=0|LM_WFA_MW=0|LM_WFA_DEFER=1" > gpurun_out/r04_c4_shard0_of_4.json 2> gpurun_out/r04_c4_shard0_of_4.err; echo "c4 rc=$?"; grep -E "A/B|index ready" gpurun_out/r04_c4_shard0_of_4.err | cut -c1-200
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_c4_shard0_of_4.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["step_ms"], d["rows"], {k:round(v) for k,v in d["stage_ms"].items()})
for k in d["kernels"][:12]: print("   ",k["name"],k["launches"],k["avg_ms"],k["exclusive_avg_ms"],k["exclusive_ms_per_step"])
print(d["ab"]); print(d["sharding_model"])
PY
