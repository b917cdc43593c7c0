#!/bin/bash
# round 3, first GPU call: parity tests (incl. the new C4/C5 shapes), the bench's own sample-row check, A/B of the anchor path
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r03_t1.log
cat gpurun_out/r03_t1.log
timeout 300 python bench.py --workload tiny --steps 2 --warmup 1 > gpurun_out/r03_tiny.json 2> gpurun_out/r03_tiny.err; echo "tiny rc=$?"
C3S="--workload c3 --genomes 20000 --families 200 --queries 2000 --steps 2 --warmup 1 --no-cpu-baseline"
timeout 600 python bench.py $C3S --tag new > gpurun_out/r03_c3s_new.json 2> gpurun_out/r03_c3s_new.err; echo "new rc=$?"
LM_PA_GLOBAL_SORT=1 timeout 600 python bench.py $C3S --tag globalsort > gpurun_out/r03_c3s_gs.json 2> gpurun_out/r03_c3s_gs.err; echo "gs rc=$?"
LM_PA_GLOBAL_SORT=1 LM_PA_SEG_BY_WAVE=1 timeout 600 python bench.py $C3S --tag old > gpurun_out/r03_c3s_old.json 2> gpurun_out/r03_c3s_old.err; echo "old rc=$?"
python - <<'PY'
import json
for t in ("new", "gs", "old"):
    try:
        p = json.loads(open("gpurun_out/r03_c3s_%s.json" % t).read().strip().split("\n")[-1])
    except Exception as e:
        print(t, "failed", e); continue
    print(t, p["value"], p["ms_per_step"], p["rows"], {k: round(v) for k, v in p["stage_ms"].items()})
    for k in p["kernels"][:14]: print("   ", k["name"], k["launches"], k["avg_ms"], k["ms_per_step"])
    for k in p["rocprim_calls"][:4]: print("   ", k["name"], k["launches"], k["avg_ms"], k["ms_per_step"])
PY
tail -3 gpurun_out/r03_tiny.err; python -c "
import json; p=json.loads(open('gpurun_out/r03_tiny.json').read().strip().split('\n')[-1]); print(p['value'], p.get('cpu_baseline'))"
