#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_two_ranks.py -m gpu -q 2>&1 | tail -3 | cut -c1-300
LM_DEBUG=1 timeout 400 python tools/loader_io_probe.py 2>&1 | grep -v "^\[lm\] \(builder\|mem\|mask\|glue\|seed image\|index resident\|scratch\)" | tail -22 | cut -c1-300 | tee gpurun_out/r05_loader_io_probe.txt
