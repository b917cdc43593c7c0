#!/bin/bash
# round 6, after the last source change (rounds no longer closed early by an idle consumer): GPU suite, the C3 passes (trace +
# three counter passes over a warm step + bench line), the C3 line as the driver runs it, the shard models while time is left
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd $(dirname $0)/.. && pwd)
cd $R; mkdir -p gpurun_out; T0=$(date +%s)
left() { echo $(( ${LIMIT:-2100} - ($(date +%s) - T0) )); }
timeout 400 python -m pytest tests -m gpu -q > gpurun_out/r06_tests_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/r06_tests_gpu.log | tail -1
bash tools/profile.sh c3 2 > gpurun_out/r06_profile_c3.log 2>&1
python bench.py --steps 10 --warmup 5 > gpurun_out/r06_c3_bench_10steps.json 2> gpurun_out/r06_c3_bench_10steps.err
for n in 8 4 2; do
  need=$(( n == 2 ? 300 : 200 ))
  if [ $(left) -gt $need ]; then timeout $need python bench.py --shard-of $n --steps 5 --warmup 5 > gpurun_out/r06_c3_shard_of_$n.json 2> gpurun_out/r06_c3_shard_of_$n.err || echo "shard-of $n cut short"; else echo "no time for shard-of $n"; fi
done
python - <<PY
import json
for f in ("c3_bench","c3_bench_10steps","c3_shard_of_2","c3_shard_of_4","c3_shard_of_8"):
    try:
        d=json.load(open("gpurun_out/r06_%s.json"%f)); m=d.get("sharding_model") or {}
        print(f, d.get("source_hash"), d["value"], d["ms_per_step"], d.get("step_ms"), (d.get("roofline") or {}).get("traffic"), {k:m.get(k) for k in ("merge_ms","predicted_step_ms","predicted_queries_per_s")})
    except Exception as e: print(f, "failed", e)
PY
