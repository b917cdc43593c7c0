#!/bin/bash
# the GPU suite and the smoke on the final tree (after the comment fixes and the tests added late in the round)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 560 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r05_tests_gpu_final.log; grep -E "passed|failed" gpurun_out/r05_tests_gpu_final.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
