#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r03_t6.log
cat gpurun_out/r03_t6.log
C3S="--workload c3 --genomes 20000 --families 200 --queries 2000 --steps 2 --warmup 1 --no-cpu-baseline"
run() { # tag, env...
  tag=$1; shift
  env "$@" timeout 600 python bench.py $C3S --tag $tag > gpurun_out/r03_c3s_$tag.json 2> gpurun_out/r03_c3s_$tag.err; echo "$tag rc=$?"
}
run pred
run nopred LM_WFA_NO_PREDICT=1
run pred_x LM_NO_PIPELINE=1 LM_WFA_SERIAL=1
timeout 600 python bench.py --workload c2 --steps 3 --warmup 1 --no-cpu-baseline --tag pred > gpurun_out/r03_c2_pred.json 2> gpurun_out/r03_c2_pred.err; echo "c2 rc=$?"
python - <<'PY'
import json
for t in ("c3s_pred", "c3s_nopred", "c3s_pred_x", "c2_pred"):
    try:
        p = json.loads(open("gpurun_out/r03_%s.json" % t).read().strip().split("\n")[-1])
    except Exception as e:
        print(t, "failed", e); continue
    print(t, p["value"], p["ms_per_step"], p["rows"], {k: round(v) for k, v in p["stage_ms"].items()}, p["work"]["wfa_retries"])
    print("   ", [(k["name"], k["launches"], k["avg_ms"], round(k["ms_per_step"])) for k in p["kernels"] if k["name"].startswith("k_wfa")])
PY
