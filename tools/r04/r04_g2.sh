#!/bin/bash
# the 2-rank spawn path of bench.py on one GPU (gloo, both ranks on device 0): index-sharded search, gather, C merge
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 100 python bench.py --gpus 1 --workload tiny --steps 2 --warmup 1 --no-cpu-baseline --no-exclusive-step > gpurun_out/r04_g1.json 2> gpurun_out/r04_g1.err; echo "g1 rc=$?"
timeout 150 python bench.py --gpus 2 --dist-backend gloo --workload tiny --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04_g2.json 2> gpurun_out/r04_g2.err; echo "g2 rc=$?"; grep -iE "error|Traceback" gpurun_out/r04_g2.err | head -5
python - <<'PY'
import json
for f in ("r04_g1","r04_g2"):
    d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1]); print(f, d["n_gpus"], d["value"], d["unit"], d["scaling"], d["rows"], d["config"]["parallelism"][:60])
PY
