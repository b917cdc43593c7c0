#!/bin/bash
# Round 5, fourth GPU call: the whole GPU suite on the cleaned-up sources (old kernel forms removed; register-ring chaining DP,
# staged partial-prefix rule of the filter, k-way split of oversize parts, genome bases streamed to the device); C3 with the
# remaining switches flipped on one resident index; the loader at C2 size; a C4 shard with the DP switch flipped.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r05_fourth_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r05_fourth_tests.log | cut -c1-300
LM_DEBUG_MEM=1 timeout 1200 python bench.py --workload c3 --steps 2 --warmup 2 --no-cpu-baseline --ab-steps 2 --ab "LM_OCC8=0|LM_PA_DP_REG=0|LM_PA_FILTER_ROLL=0" > gpurun_out/r05_c3_ab4.json 2> gpurun_out/r05_c3_ab4.err; echo "c3 rc=$?"; grep -E "A/B|halved|lane slabs:" gpurun_out/r05_c3_ab4.err | cut -c1-250 | head -12
LM_DEBUG=1 timeout 600 python bench.py --workload c2 --steps 1 --warmup 0 --no-cpu-baseline --no-exclusive-step --loader-check > gpurun_out/r05_c2_loader2.json 2> gpurun_out/r05_c2_loader2.err; echo "loader rc=$?"; grep -E "loader check|loader:" gpurun_out/r05_c2_loader2.err | cut -c1-330 | tail -6
timeout 600 python bench.py --workload c4 --steps 2 --warmup 3 --shard-rank 0 --ab-steps 2 --ab "LM_PA_DP_REG=0|LM_OCC8=0" > gpurun_out/r05_c4_shard0_ab2.json 2> gpurun_out/r05_c4_shard0_ab2.err; echo "c4 rc=$?"
python - <<'PY'
import json
for f in ("r05_c3_ab4", "r05_c2_loader2", "r05_c4_shard0_ab2"):
    try:
        d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], "first", d.get("first_step_ms"), d.get("warmup_step_ms"), d.get("step_ms"), "rows", d["rows"])
        print("   ab", d.get("ab"), "loader", (d.get("loader") or {}).get("open_s"), (d.get("loader") or {}).get("open_GBps_of_files"))
        print("   stage_ms", d["stage_ms"])
        rp = d["roofline_pipeline"]
        print("   kernel ms/step", rp["kernel_ms_per_step"], "exclusive", rp["exclusive_kernel_ms_per_step"])
        for k in d["kernels"][:12]:
            print("    %-22s launches %6d avg %9.3f ms/step %9.1f excl/step %s" % (k["name"], k["launches"], k["avg_ms"], k["ms_per_step"], k["exclusive_ms_per_step"]))
    except Exception as e:
        print(f, "no line:", e)
PY
