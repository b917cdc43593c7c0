#!/bin/bash
# SQ instruction-mix counters for the search kernels (one PMC pass, no tracing), run through gpurun
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_sq
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/prof_sq -o sq -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /tmp/prof_sq.log 2>&1
tail -3 /tmp/prof_sq.log
python $R/tools/summarize_rocprof.py /tmp/prof_sq $R/gpurun_out/r01_c2_pmc_sq.json | grep -E "^k_(wfa|pa_|extend|lookup)"
