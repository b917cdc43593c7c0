#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r03_tests_gpu.log
cat gpurun_out/r03_tests_gpu.log
bash tools/profile.sh c2 3 2>&1 | tail -30
bash tools/profile.sh c3 3 2>&1 | tail -40
