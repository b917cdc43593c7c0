#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2; do
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -14 > gpurun_out/r04_tests_gpu_h$i.log; tail -3 gpurun_out/r04_tests_gpu_h$i.log | head -2
done
