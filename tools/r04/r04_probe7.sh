#!/bin/bash
# the whole GPU suite (incl. the shim binary against the reference's golden TSVs), then the flat anchor emit on / off
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r04_tests_gpu_d.log; tail -6 gpurun_out/r04_tests_gpu_d.log
C3S="--workload c3 --genomes 20000 --queries 2000 --families 200 --steps 2 --warmup 1 --no-cpu-baseline"
for F in 1 0; do
LM_LOOKUP_FLAT=$F timeout 600 python bench.py $C3S --tag flat$F > gpurun_out/r04_c3s_p7_flat$F.json 2> gpurun_out/r04_c3s_p7_flat$F.err; echo "flat$F rc=$?"
done
python - <<'PY'
import json
for t in ("flat1","flat0"):
    d=json.loads(open("gpurun_out/r04_c3s_p7_%s.json"%t).read().strip().splitlines()[-1])
    print(t, d["value"], d["ms_per_step"], d["step_ms"], d["rows"], d["stage_ms"]["ms_lookup"])
    for k in d["kernels"]+d["rocprim_calls"]:
        if "lookup" in k["name"] or k["name"] in ("scan","sort_anchors"): print("   ",k["name"],k["launches"],k["avg_ms"],k["exclusive_avg_ms"],k["achieved_GBs"])
    print("   ", {k:d["roofline_seed_lookup"][k] for k in ("stage","stage_ms","stage_frac","frac","achieved")})
PY
