#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel stats of the bench command, separate PMC passes (FETCH_SIZE, WRITE_SIZE,
# SQ instruction mix / waits) - counters are never collected together with tracing - each over ONE WARM step: bench.py puts an
# lm::k_profile_mark kernel in front of and behind its timed steps and tools/summarize_rocprof.py keeps the dispatches between
# them (rounds 1-5 counted the cold first step of a fresh handle) - and LAST the bench line of the workload,
# which reads the summaries of those passes (copied into profiles/ of the box's copy of the tree first), so that
# roofline.traffic / instruction_issue of the line are filled by the run itself and nothing is refilled afterwards.
# Every summary is stamped with the hash of the library sources (bench.py source_hash): bench.py refuses PMC passes taken
# on other sources.   usage: tools/profile.sh <workload> [steps] [extra bench args...]
P=${LM_PROFILE_PREFIX:-r06}
W=${1:-c3}
STEPS=${2:-2}
shift; shift
EXTRA="$@"
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd $R
H=$(python -c "import bench; print(bench.source_hash())")
echo "workload $W, sources $H"
cd /tmp
rm -rf /tmp/prof_ks /tmp/prof_f /tmp/prof_w /tmp/prof_sq
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ks -o ks -- python $R/bench.py --workload $W --steps $STEPS --warmup 2 --no-cpu-baseline --no-exclusive-step $EXTRA > /tmp/prof_ks.log 2>&1
python $R/tools/summarize_rocprof.py /tmp/prof_ks $R/gpurun_out/${P}_${W}_kernel_stats.json $H "rocprofv3 --kernel-trace --stats -- python bench.py --workload $W --steps $STEPS --warmup 2 --no-cpu-baseline --no-exclusive-step $EXTRA"
for f in $(find /tmp/prof_ks -name "*kernel_stats.csv"); do python - "$f" "$R/gpurun_out/${P}_${W}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
w = csv.writer(open(sys.argv[2], "w"))
for r in rows:
    r[0] = r[0][:100]   # rocPRIM template names are kilobytes long
    w.writerow(r)
PY
done
timeout 1500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f -o f -- python $R/bench.py --workload $W --steps 1 --warmup 2 --no-cpu-baseline --no-exclusive-step $EXTRA > /tmp/prof_f.log 2>&1
python $R/tools/summarize_rocprof.py /tmp/prof_f $R/gpurun_out/${P}_${W}_pmc_fetch.json $H "rocprofv3 --pmc FETCH_SIZE -- python bench.py --workload $W --steps 1 --warmup 2 --no-cpu-baseline --no-exclusive-step $EXTRA" | tail -12
timeout 1500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w -o w -- python $R/bench.py --workload $W --steps 1 --warmup 2 --no-cpu-baseline --no-exclusive-step $EXTRA > /tmp/prof_w.log 2>&1
python $R/tools/summarize_rocprof.py /tmp/prof_w $R/gpurun_out/${P}_${W}_pmc_write.json $H "rocprofv3 --pmc WRITE_SIZE -- python bench.py --workload $W --steps 1 --warmup 2 --no-cpu-baseline --no-exclusive-step $EXTRA" | tail -12
timeout 1500 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/prof_sq -o sq -- python $R/bench.py --workload $W --steps 1 --warmup 2 --no-cpu-baseline --no-exclusive-step $EXTRA > /tmp/prof_sq.log 2>&1
python $R/tools/summarize_rocprof.py /tmp/prof_sq $R/gpurun_out/${P}_${W}_pmc_sq.json $H "rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- python bench.py --workload $W --steps 1 --warmup 2 --no-cpu-baseline --no-exclusive-step $EXTRA" | grep -E "^k_(wfa|pa_|extend|lookup|chain)"
tail -n 2 /tmp/prof_f.log /tmp/prof_w.log /tmp/prof_sq.log
# the passes of THESE sources where bench.py looks for them, then the bench line
cp $R/gpurun_out/${P}_${W}_pmc_fetch.json $R/gpurun_out/${P}_${W}_pmc_write.json $R/gpurun_out/${P}_${W}_pmc_sq.json $R/profiles/
cd $R
timeout 1500 python bench.py --workload $W --steps 3 --warmup 3 $EXTRA > gpurun_out/${P}_${W}_bench.json 2> gpurun_out/${P}_${W}_bench.err
tail -2 gpurun_out/${P}_${W}_bench.err
