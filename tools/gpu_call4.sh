#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r03_t4.log
cat gpurun_out/r03_t4.log
C3S="--workload c3 --genomes 20000 --families 200 --queries 2000 --steps 2 --warmup 1 --no-cpu-baseline"
run() { # tag, env...
  tag=$1; shift
  env "$@" timeout 600 python bench.py $C3S --tag $tag > gpurun_out/r03_c3s_$tag.json 2> gpurun_out/r03_c3s_$tag.err; echo "$tag rc=$?"
}
run w00101
run w00111 LM_WFA_WIN=00111
run w00001 LM_WFA_WIN=00001
run w01101 LM_WFA_WIN=01101
run x00101 LM_NO_PIPELINE=1 LM_WFA_SERIAL=1
run x00111 LM_NO_PIPELINE=1 LM_WFA_SERIAL=1 LM_WFA_WIN=00111
python - <<'PY'
import json
for t in ("w00101", "w00111", "w00001", "w01101", "x00101", "x00111"):
    try:
        p = json.loads(open("gpurun_out/r03_c3s_%s.json" % t).read().strip().split("\n")[-1])
    except Exception as e:
        print(t, "failed", e); continue
    print(t, p["value"], p["ms_per_step"], p["rows"], {k: round(v) for k, v in p["stage_ms"].items()}, p["work"]["wfa_retries"])
    print("   ", [(k["name"], k["launches"], k["avg_ms"], round(k["ms_per_step"])) for k in p["kernels"] if k["name"].startswith("k_wfa")])
PY
