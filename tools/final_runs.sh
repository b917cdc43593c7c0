#!/bin/bash
# the round-end batch on the GPU box (one gpurun call): parity tests, the C3 profile set, the C4 / C5 shards, the shard-of-8 run, the C2 profile set
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r04_tests_gpu.log
cat gpurun_out/r04_tests_gpu.log
bash tools/profile.sh c3 3 2>&1 | grep -E "^workload|rank 0|^k_wfa_lean |^k_pa_search" | head -12
timeout 400 python bench.py --workload c4 --steps 2 --warmup 1 --shard-rank 0 > gpurun_out/r04_c4_shard0_of_4.json 2> gpurun_out/r04_c4_shard0_of_4.err; echo "c4 rc=$?"; tail -2 gpurun_out/r04_c4_shard0_of_4.err
timeout 400 python bench.py --workload c5 --steps 2 --warmup 1 --shard-rank 0 > gpurun_out/r04_c5_shard0_of_8.json 2> gpurun_out/r04_c5_shard0_of_8.err; echo "c5 rc=$?"; tail -2 gpurun_out/r04_c5_shard0_of_8.err
timeout 200 python bench.py --workload c3 --steps 2 --warmup 1 --shard-of 8 --shard-rank 1 --no-exclusive-step > gpurun_out/r04_c3_shard_of_8.json 2> gpurun_out/r04_c3_shard_of_8.err; echo "shard-of 8 rc=$?"
bash tools/profile.sh c2 3 2>&1 | grep -E "^workload|rank 0" | head -4
