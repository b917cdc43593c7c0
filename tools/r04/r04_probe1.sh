#!/bin/bash
# round-4 first GPU call: parity tests on the adopted kernels, then the c3-shaped development workload (2 000 reads vs
# 20 000 x 2-Mb genomes) with each new kernel on / off, the per-wavefront timing of k_wfa_lean, and the full-index oracle tie
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04_tests_gpu_a.log; tail -3 gpurun_out/r04_tests_gpu_a.log
C3S="--workload c3 --genomes 20000 --queries 2000 --families 200 --steps 2 --warmup 1"
timeout 600 python bench.py $C3S --no-cpu-baseline --tag mw1 > gpurun_out/r04_c3s_mw1.json 2> gpurun_out/r04_c3s_mw1.err; echo "mw1 rc=$?"
LM_WFA_MW=0 timeout 600 python bench.py $C3S --no-cpu-baseline --tag mw0 > gpurun_out/r04_c3s_mw0.json 2> gpurun_out/r04_c3s_mw0.err; echo "mw0 rc=$?"
LM_PA_CHAIN_RING=0 timeout 600 python bench.py $C3S --no-cpu-baseline --no-exclusive-step --tag ring0 > gpurun_out/r04_c3s_ring0.json 2> gpurun_out/r04_c3s_ring0.err; echo "ring0 rc=$?"
LM_DEBUG_WFA_WAVES=gpurun_out/r04_c3s_waves.jsonl timeout 600 python bench.py $C3S --no-cpu-baseline --tag waves > gpurun_out/r04_c3s_waves.json 2> gpurun_out/r04_c3s_waves.err; echo "waves rc=$?"
timeout 900 python bench.py $C3S --tag full > gpurun_out/r04_c3s_full.json 2> gpurun_out/r04_c3s_full.err; echo "full rc=$?"; tail -3 gpurun_out/r04_c3s_full.err
python - <<'PY'
import json
for t in ("mw1","mw0","ring0","waves","full"):
    try:
        d=json.loads(open("gpurun_out/r04_c3s_%s.json"%t).read().strip().splitlines()[-1])
        ks={k["name"]:(k["exclusive_ms_per_step"] or k["ms_per_step"]) for k in d["kernels"]}
        print(t, d["value"], d["ms_per_step"], d["rows"], {k:v for k,v in ks.items() if k.startswith(("k_wfa","k_pa_chain"))})
        if t=="full": print(json.dumps(d["cpu_baseline"])[:1500])
    except Exception as e: print(t,"failed",e)
PY
