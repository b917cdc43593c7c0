#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
LM_DEBUG=1 timeout 900 python bench.py --workload c2 --genomes 2500 --families 25 --steps 1 --warmup 0 --no-cpu-baseline --no-exclusive-step --loader-check --tag loader_q > gpurun_out/r04_c2q_loader.json 2> gpurun_out/r04_c2q_loader.err; echo "rc=$?"; grep -E "loader check|loader:" gpurun_out/r04_c2q_loader.err | cut -c1-330
