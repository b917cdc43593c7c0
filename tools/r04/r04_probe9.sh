#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_gather.py tests/test_gpu_megabase.py tests/test_gpu_builder.py tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -12 > gpurun_out/r04_tests_gpu_f.log; tail -4 gpurun_out/r04_tests_gpu_f.log
timeout 900 python bench.py --workload c2 --genomes 2500 --families 25 --steps 1 --warmup 1 --no-cpu-baseline --no-exclusive-step --loader-check --tag loader_quarter > gpurun_out/r04_c2q_loader.json 2> gpurun_out/r04_c2q_loader.err; echo "rc=$?"; grep -E "loader check|index ready" gpurun_out/r04_c2q_loader.err | cut -c1-700
