#!/bin/bash
# Round 5, first GPU call (via gpurun, AFTER `python tools/adopt_wfa_lean2.py` and a rebuild here: built .so files travel):
#   gpurun --timeout 2400 -- 'bash tools/r05_first.sh'
# 1. the forced-path check of k_wfa_lean2 / k_wfa_mw2 against the oracle with the switch on and off (+ its two timing batches);
# 2. the GPU tests that touch the WFA kernels;  3. C2 and C3 with the same resident index under LM_WFA_LEAN2=0 (A/B).
# What to look at: gpurun_out/r05_wfa_lean2_check.json ("different": 0 everywhere, "seconds": lean2=1 vs 0), the test tail,
# the "ab" object of the two bench lines.  Counter passes / profiles (tools/profile.sh) only after this is green.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python experiments/wfa_lean2/gpu_check.py > gpurun_out/r05_wfa_lean2_check.log 2>&1; echo "gpu_check rc=$?"; tail -5 gpurun_out/r05_wfa_lean2_check.log
timeout 1200 python -m pytest tests/test_gpu_wfa_lean2.py tests/test_gpu_wfa_mw.py tests/test_gpu_longreads.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r05_first_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r05_first_tests.log
timeout 600 python bench.py --workload c2 --steps 5 --warmup 2 --no-cpu-baseline --ab-steps 3 --ab "LM_WFA_LEAN2=0" > gpurun_out/r05_c2_ab.json 2> gpurun_out/r05_c2_ab.err; echo "c2 rc=$?"
timeout 1500 python bench.py --workload c3 --steps 2 --warmup 3 --no-cpu-baseline --ab-steps 2 --ab "LM_WFA_LEAN2=0" > gpurun_out/r05_c3_ab.json 2> gpurun_out/r05_c3_ab.err; echo "c3 rc=$?"
python - <<'PY'
import json
for w in ("c2", "c3"):
    try:
        d = json.loads(open("gpurun_out/r05_%s_ab.json" % w).read().strip().splitlines()[-1])
        print(w, d["value"], d["ms_per_step"], d.get("step_ms"), d.get("ab"))
        for k in d["kernels"][:8]:
            print("   ", k["name"], k["launches"], k["avg_ms"], k["ms_per_step"])
    except Exception as e:
        print(w, "no line:", e)
PY
