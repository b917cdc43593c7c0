#!/bin/bash
# exclusive kernel timings (no pipelining, WFA classes one after the other) of the three anchor-path variants + L2 traffic
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
export LM_NO_PIPELINE=1 LM_WFA_SERIAL=1
C3S="--workload c3 --genomes 20000 --families 200 --queries 2000 --steps 2 --warmup 1 --no-cpu-baseline"
timeout 600 python bench.py $C3S --tag new_x > gpurun_out/r03_c3s_new.json 2> gpurun_out/r03_c3s_new.err; echo "new rc=$?"
LM_PA_GLOBAL_SORT=1 timeout 600 python bench.py $C3S --tag globalsort_x > gpurun_out/r03_c3s_gs.json 2> gpurun_out/r03_c3s_gs.err; echo "gs rc=$?"
LM_PA_GLOBAL_SORT=1 LM_PA_SEG_BY_WAVE=1 timeout 600 python bench.py $C3S --tag old_x > gpurun_out/r03_c3s_old.json 2> gpurun_out/r03_c3s_old.err; echo "old rc=$?"
python - <<'PY'
import json
for t in ("new", "gs", "old"):
    try:
        p = json.loads(open("gpurun_out/r03_c3s_%s.json" % t).read().strip().split("\n")[-1])
    except Exception as e:
        print(t, "failed", e); continue
    print(t, p["value"], p["ms_per_step"], p["rows"], {k: round(v) for k, v in p["stage_ms"].items()})
    for k in p["kernels"][:16]: print("   ", k["name"], k["launches"], k["avg_ms"], k["ms_per_step"])
    for k in p["rocprim_calls"][:4]: print("   ", k["name"], k["launches"], k["avg_ms"], k["ms_per_step"])
PY
cd /tmp
H=x
for v in new old; do
  rm -rf /tmp/prof_f
  if [ $v = old ]; then export LM_PA_GLOBAL_SORT=1 LM_PA_SEG_BY_WAVE=1; fi
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f -o f -- python $GRAFT_REPO_ROOT/bench.py --workload c3 --genomes 20000 --families 200 --queries 2000 --steps 1 --warmup 0 --no-cpu-baseline > /tmp/prof_f.log 2>&1
  python $GRAFT_REPO_ROOT/tools/summarize_rocprof.py /tmp/prof_f $GRAFT_REPO_ROOT/gpurun_out/r03_c3s_${v}_pmc_fetch.json $H "fetch $v" | grep -E "^k_pa|sort"
done
