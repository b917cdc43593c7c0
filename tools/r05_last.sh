#!/bin/bash
# the loader at C2 size after the genome runs reach the device in one copy each (lm_format.cpp only)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
LM_DEBUG=1 timeout 165 python bench.py --workload c2 --steps 1 --warmup 0 --no-cpu-baseline --no-exclusive-step --loader-check > gpurun_out/r05_c2_loader_packed.json 2> gpurun_out/r05_c2_loader_packed.err; echo "loader rc=$?"; grep -E "loader check|loader:" gpurun_out/r05_c2_loader_packed.err | cut -c1-420 | tail -6
