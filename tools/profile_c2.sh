#!/bin/bash
# Runs on the GPU box (via gpurun): default bench line, rocprofv3 kernel stats, and two separate PMC passes.
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd $R
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err
tail -3 gpurun_out/bench_c2.err
cd /tmp
rm -rf /tmp/prof_ks /tmp/prof_f /tmp/prof_w
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ks -o c2 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /tmp/prof_ks.log 2>&1
python $R/tools/summarize_rocprof.py /tmp/prof_ks $R/gpurun_out/r01_c2_kernel_stats.json
for f in $(find /tmp/prof_ks -name "*kernel_stats.csv"); do python - "$f" "$R/gpurun_out/r01_c2_kernel_stats.csv" <<'PY'
import csv, sys, re
rows = list(csv.reader(open(sys.argv[1])))
w = csv.writer(open(sys.argv[2], "w"))
for r in rows:
    r[0] = r[0][:100]   # rocPRIM template names are kilobytes long
    w.writerow(r)
PY
done
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f -o c2f -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /tmp/prof_f.log 2>&1
python $R/tools/summarize_rocprof.py /tmp/prof_f $R/gpurun_out/r01_c2_pmc_fetch.json
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w -o c2w -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /tmp/prof_w.log 2>&1
python $R/tools/summarize_rocprof.py /tmp/prof_w $R/gpurun_out/r01_c2_pmc_write.json
tail -n 2 /tmp/prof_f.log; tail -n 2 /tmp/prof_w.log
