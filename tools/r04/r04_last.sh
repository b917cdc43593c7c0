#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 150 python bench.py --workload c3 --steps 3 --warmup 3 --shard-of 8 --shard-rank 1 --no-exclusive-step > gpurun_out/r04_c3_shard_of_8_w3.json 2> gpurun_out/r04_c3_shard_of_8_w3.err; echo "shard-of 8 rc=$?"
timeout 200 python bench.py --workload c4 --steps 2 --warmup 3 --shard-rank 0 --no-exclusive-step --ab-steps 2 --ab "LM_TWO_LANES=0" > gpurun_out/r04_c4_shard0_of_4_w3.json 2> gpurun_out/r04_c4_shard0_of_4_w3.err; echo "c4 rc=$?"; grep "A/B" gpurun_out/r04_c4_shard0_of_4_w3.err
python - <<'PY'
import json
for f in ("r04_c3_shard_of_8_w3","r04_c4_shard0_of_4_w3"):
    try:
        d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["step_ms"], d["rows"], d["sharding_model"]["predicted_queries_per_s"], d["sharding_model"]["merge_ms_host"], d.get("ab"))
    except Exception as e: print(f,"failed",e)
PY
