#!/bin/bash
# Round 5, first GPU call:  gpurun --timeout 2400 -- 'bash tools/r05_first.sh'
# The kernels staged in round 4 are product code now (LM_WFA_LEAN2, LM_PA_CHAIN_BT_WAVE, LM_PA_CHAIN_PIPE, lane slabs): this
# is their first run on a GPU.  1. forced-path check of k_wfa_lean2 / k_wfa_mw2 against the oracle, switch on and off (+ two
# timing batches);  2. the GPU tests that touch what changed;  3. C3 on one resident index: cold step, steady steps, the
# serialised step, then each switch flipped (A/B);  4. phase clocks of the chaining / search kernels on a c3-shaped index;
# 5. C2 with A/B.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tests/wfa_lean2_gpu_check.py > gpurun_out/r05_wfa_lean2_check.log 2>&1; echo "gpu_check rc=$?"; grep -c DIFFERENT gpurun_out/r05_wfa_lean2_check.log; tail -12 gpurun_out/r05_wfa_lean2_check.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_wfa_lean2.py tests/test_gpu_wfa_mw.py tests/test_gpu_longreads.py tests/test_gpu_parity.py tests/test_gpu_c4c5.py -m gpu -x -q > gpurun_out/r05_first_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r05_first_tests.log | cut -c1-300
LM_DEBUG_MEM=1 timeout 1200 python bench.py --workload c3 --steps 2 --warmup 2 --no-cpu-baseline --ab-steps 2 --ab "LM_WFA_LEAN2=0|LM_PA_CHAIN_BT_WAVE=0|LM_PA_CHAIN_PIPE=0|LM_WFA_LEAN2=0 LM_PA_CHAIN_BT_WAVE=0 LM_PA_CHAIN_PIPE=0" > gpurun_out/r05_c3_ab.json 2> gpurun_out/r05_c3_ab.err; echo "c3 rc=$?"; grep -E "A/B|scratch|slab" gpurun_out/r05_c3_ab.err | cut -c1-250 | head -20
LM_TWO_LANES=0 LM_DEBUG_PA_CHAIN=1 LM_DEBUG_PA_SEARCH=1 timeout 400 python bench.py --workload c3 --genomes 20000 --queries 2000 --families 201 --steps 1 --warmup 1 --no-cpu-baseline --no-exclusive-step > gpurun_out/r05_c3s_clocks.json 2> gpurun_out/r05_c3s_clocks.err; echo "c3s clocks rc=$?"
python - <<'PY'
import re, collections
acc = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0, 0, 0])
for ln in open("gpurun_out/r05_c3s_clocks.err"):
    m = re.search(r"k_pa_chain: (\d+) windows \((\d+) finished by the wavefront kernel, (\d+) handed.*clear\+trim ([\d.]+), DP ([\d.]+), backtrack ([\d.]+)", ln)
    if m:
        a = acc["chain"]; a[0] += 1; a[1] += float(m.group(4)); a[2] += float(m.group(5)); a[3] += float(m.group(6)); a[4] += int(m.group(1)); a[5] += int(m.group(3))
    m = re.search(r"k_pa_search: (\d+) candidates in (\d+) wavefront passes.*?(\d+) enumeration passes, (\d+) anchors.*search ([\d.]+), enumeration ([\d.]+)", ln)
    if m:
        a = acc["search"]; a[0] += 1; a[1] += float(m.group(5)); a[2] += float(m.group(6)); a[4] += int(m.group(1)); a[5] += int(m.group(4)); a[3] += int(m.group(2))
print("k_pa_chain launches %d: wavefront-ms clear+trim %.0f DP %.0f backtrack %.0f; windows %d, to the workgroup kernel %d" % tuple(acc["chain"]))
print("k_pa_search launches %d: wavefront-ms search %.0f enumeration %.0f; passes %d, candidates %d, anchors %d" % tuple(acc["search"]))
PY
timeout 500 python bench.py --workload c2 --steps 5 --warmup 2 --no-cpu-baseline --ab-steps 3 --ab "LM_WFA_LEAN2=0|LM_PA_CHAIN_BT_WAVE=0" > gpurun_out/r05_c2_ab.json 2> gpurun_out/r05_c2_ab.err; echo "c2 rc=$?"
python - <<'PY'
import json
for w in ("c3", "c3s_clocks", "c2"):
    try:
        d = json.loads(open("gpurun_out/r05_%s%s.json" % (w, "" if w == "c3s_clocks" else "_ab")).read().strip().splitlines()[-1])
        print(w, d["value"], d["ms_per_step"], "first", d.get("first_step_ms"), d.get("warmup_step_ms"), d.get("step_ms"), "rows", d["rows"])
        print("   ab", d.get("ab"))
        print("   stage_ms", d["stage_ms"])
        rp = d["roofline_pipeline"]
        print("   kernel ms/step", rp["kernel_ms_per_step"], "exclusive", rp["exclusive_kernel_ms_per_step"])
        for k in d["kernels"][:14]:
            print("    %-22s launches %6d avg %9.3f ms/step %9.1f excl/step %s" % (k["name"], k["launches"], k["avg_ms"], k["ms_per_step"], k["exclusive_ms_per_step"]))
    except Exception as e:
        print(w, "no line:", e)
PY
