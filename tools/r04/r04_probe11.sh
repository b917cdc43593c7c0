#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r04_tests_gpu_g.log; tail -4 gpurun_out/r04_tests_gpu_g.log
LM_DEBUG=1 timeout 900 python bench.py --workload c2 --genomes 2500 --families 25 --steps 1 --warmup 0 --no-cpu-baseline --no-exclusive-step --loader-check --tag loader_q > gpurun_out/r04_c2q_loader.json 2> gpurun_out/r04_c2q_loader.err; echo "rc=$?"; grep -E "loader check|loader:" gpurun_out/r04_c2q_loader.err | cut -c1-330
