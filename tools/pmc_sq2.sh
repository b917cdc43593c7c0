#!/bin/bash
# SQ wait/active counters for the search kernels (one PMC pass, no tracing), run through gpurun
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_sq2
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d /tmp/prof_sq2 -o sq2 -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /tmp/prof_sq2.log 2>&1
tail -n 3 /tmp/prof_sq2.log
python $R/tools/summarize_rocprof.py /tmp/prof_sq2 $R/gpurun_out/r01_c2_pmc_sq2.json | grep -E "^k_(wfa_lean|pa_anchors|pa_chain|extend |lookup_count)"
