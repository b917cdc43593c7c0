#!/bin/bash
# Round 5, third GPU call: the whole GPU suite; the cold C3 step with 95 % of the budget in lane slabs; the loader with decoded
# seeds kept on the device between its passes (C2 size); a C4 shard with the chaining switches flipped; one shard of the
# 8-GPU C3 run in steady state.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r05_third_tests.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r05_third_tests.log | cut -c1-300
LM_DEBUG_MEM=1 timeout 900 python bench.py --workload c3 --steps 2 --warmup 2 --no-cpu-baseline --no-exclusive-step > gpurun_out/r05_c3_cold.json 2> gpurun_out/r05_c3_cold.err; echo "c3 rc=$?"; grep -E "halved|lane slabs" gpurun_out/r05_c3_cold.err | cut -c1-330 | head -12
LM_DEBUG=1 timeout 600 python bench.py --workload c2 --steps 1 --warmup 1 --no-cpu-baseline --no-exclusive-step --loader-check > gpurun_out/r05_c2_loader.json 2> gpurun_out/r05_c2_loader.err; echo "loader rc=$?"; grep -E "loader check|loader:" gpurun_out/r05_c2_loader.err | cut -c1-400 | tail -8
timeout 600 python bench.py --workload c4 --steps 2 --warmup 3 --shard-rank 0 --ab-steps 2 --ab "LM_PA_CHAIN_PIPE=0|LM_PA_CHAIN_BT_WAVE=0|LM_TWO_LANES=0" > gpurun_out/r05_c4_shard0_of_4.json 2> gpurun_out/r05_c4_shard0_of_4.err; echo "c4 rc=$?"
timeout 400 python bench.py --workload c3 --steps 3 --warmup 3 --shard-of 8 --shard-rank 1 --no-exclusive-step > gpurun_out/r05_c3_shard_of_8.json 2> gpurun_out/r05_c3_shard_of_8.err; echo "shard-of 8 rc=$?"
python - <<'PY'
import json
for f in ("r05_c3_cold", "r05_c2_loader", "r05_c4_shard0_of_4", "r05_c3_shard_of_8"):
    try:
        d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], "first", d.get("first_step_ms"), d.get("warmup_step_ms"), d.get("step_ms"), "rows", d["rows"])
        print("   ab", d.get("ab"), "loader", d.get("loader"))
        print("   stage_ms", d["stage_ms"])
        print("   model", (d.get("sharding_model") or {}))
        for k in d["kernels"][:8]:
            print("    %-22s launches %6d avg %9.3f ms/step %9.1f excl/step %s" % (k["name"], k["launches"], k["avg_ms"], k["ms_per_step"], k["exclusive_ms_per_step"]))
    except Exception as e:
        print(f, "no line:", e)
PY
