#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_gpu_gather.py tests/test_gpu_gather_c.py tests/test_gpu_two_ranks.py tests/test_gpu_shim.py -m gpu -q -x 2>&1 | grep -v "RCCL\|HIP ver\|ROCm ver\|Hostname\|Librccl" | tail -25 | cut -c1-300
