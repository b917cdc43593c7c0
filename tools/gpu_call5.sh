#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
C3S="--workload c3 --genomes 20000 --families 200 --queries 2000 --steps 2 --warmup 1 --no-cpu-baseline"
run() { # tag, env...
  tag=$1; shift
  env "$@" timeout 600 python bench.py $C3S --tag $tag > gpurun_out/r03_c3s_$tag.json 2> gpurun_out/r03_c3s_$tag.err; echo "$tag rc=$?"
}
run r700k LM_DEBUG_ROUND_HSPS=700000 LM_DEBUG_MIN_ROUND_HSPS=300000
run r1m LM_DEBUG_ROUND_HSPS=1000000 LM_DEBUG_MIN_ROUND_HSPS=500000
run nc16 LM_WFA_FIRST_NC=2,2,4,16,16
run dump LM_DEBUG_WFA_DUMP=/tmp/wfa_dump.txt
python - <<'PY'
import json
for t in ("r700k", "r1m", "nc16", "dump"):
    try:
        p = json.loads(open("gpurun_out/r03_c3s_%s.json" % t).read().strip().split("\n")[-1])
    except Exception as e:
        print(t, "failed", e); continue
    print(t, p["value"], p["ms_per_step"], p["rows"], {k: round(v) for k, v in p["stage_ms"].items()}, p["work"]["wfa_retries"])
    print("   ", [(k["name"], k["launches"], k["avg_ms"], round(k["ms_per_step"])) for k in p["kernels"] if k["name"].startswith("k_wfa")])
PY
# keep the dump of ONE step (it is appended per pass over 3 steps): classes >= 256 diagonals only
awk '$1 >= 256' /tmp/wfa_dump.txt | sort -u | head -200000 > gpurun_out/r03_wfa_dump_wide.txt
awk '$1 == 128 && $2 == 3' /tmp/wfa_dump.txt | sort -u | head -50000 > gpurun_out/r03_wfa_dump_128fail.txt
wc -l /tmp/wfa_dump.txt gpurun_out/r03_wfa_dump_wide.txt gpurun_out/r03_wfa_dump_128fail.txt
