#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r03_t11.log
cat gpurun_out/r03_t11.log
C3S="--workload c3 --genomes 20000 --families 200 --queries 2000 --steps 2 --warmup 1 --no-cpu-baseline --no-exclusive-step"
run() { tag=$1; shift; env "$@" timeout 200 python bench.py $C3S --tag $tag > gpurun_out/r03_c3s_$tag.json 2> gpurun_out/r03_c3s_$tag.err; echo "$tag rc=$?"; }
run p100
run p75 LM_WFA_RESIDENT_PCT=75
run p50 LM_WFA_RESIDENT_PCT=50
run p75l2 LM_WFA_RESIDENT_PCT=75 LM_TWO_LANES=1
python - <<'PY'
import json
for t in ("p100", "p75", "p50", "p75l2"):
    try:
        p = json.loads(open("gpurun_out/r03_c3s_%s.json" % t).read().strip().split("\n")[-1])
    except Exception as e:
        print(t, "failed", e); continue
    print(t, p["value"], p["ms_per_step"], p["rows"], {k: round(v) for k, v in p["stage_ms"].items()})
    print("   ", [(k["name"], k["launches"], k["avg_ms"]) for k in p["kernels"][:9]])
PY
