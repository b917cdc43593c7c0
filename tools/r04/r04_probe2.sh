#!/bin/bash
# timeline of one c3-shaped step (LM_DEBUG stamps) with the mw kernels at raised priority
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
C3S="--workload c3 --genomes 20000 --queries 2000 --families 200 --steps 2 --warmup 1"
timeout 600 python bench.py $C3S --no-cpu-baseline --tag mwprio > gpurun_out/r04_c3s_mwprio.json 2> gpurun_out/r04_c3s_mwprio.err; echo "mwprio rc=$?"
LM_DEBUG=1 timeout 600 python bench.py --workload c3 --genomes 20000 --queries 2000 --families 200 --steps 1 --warmup 1 --no-cpu-baseline --no-exclusive-step --tag timeline > gpurun_out/r04_c3s_timeline.json 2> gpurun_out/r04_c3s_timeline.err; echo "timeline rc=$?"
grep -E "^\[lm \+" gpurun_out/r04_c3s_timeline.err > gpurun_out/r04_c3s_timeline.txt; wc -l gpurun_out/r04_c3s_timeline.txt
python - <<'PY'
import json
for t in ("mwprio",):
    d=json.loads(open("gpurun_out/r04_c3s_%s.json"%t).read().strip().splitlines()[-1])
    print(t, d["value"], d["ms_per_step"], d["rows"], d["stage_ms"])
    for k in d["kernels"]:
        if k["name"].startswith("k_wfa"): print(k["name"],k["launches"],k["avg_ms"],k["exclusive_avg_ms"])
PY
