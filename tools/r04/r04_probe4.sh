#!/bin/bash
# do the stalls of the predicted-width chains come from streams sharing hardware queues?  (ROCclr maps streams onto GPU_MAX_HW_QUEUES = 4 queues)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
C3S="--workload c3 --genomes 20000 --queries 2000 --families 200 --steps 3 --warmup 1 --no-cpu-baseline --no-exclusive-step"
for Q in 4 8 16 24; do
GPU_MAX_HW_QUEUES=$Q timeout 600 python bench.py $C3S --tag q$Q > gpurun_out/r04_c3s_p4_q$Q.json 2> gpurun_out/r04_c3s_p4_q$Q.err; echo "q$Q rc=$?"
GPU_MAX_HW_QUEUES=$Q LM_WFA_AK_MARGIN=-1 timeout 600 python bench.py $C3S --tag q${Q}_noak > gpurun_out/r04_c3s_p4_q${Q}_noak.json 2> gpurun_out/r04_c3s_p4_q${Q}_noak.err; echo "q${Q}_noak rc=$?"
done
python - <<'PY'
import json
for t in ("q4","q4_noak","q8","q8_noak","q16","q16_noak","q24","q24_noak"):
    try:
        d=json.loads(open("gpurun_out/r04_c3s_p4_%s.json"%t).read().strip().splitlines()[-1])
        print(t, d["value"], d["ms_per_step"], d["rows"], {k:round(v) for k,v in d["stage_ms"].items()})
    except Exception as e: print(t,"failed",e)
PY
