#!/bin/bash
# Round 5, second GPU call: the rolled k_pa_filter + 24-bit Bloom hash, shrink margins of k_wfa_lean2, lane slabs cut at open,
# lm_gather_rows; the whole GPU suite, C3 A/B on one resident index, size-class clocks of k_pa_chain, an SQ pass at C2.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05_second_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r05_second_tests.log | cut -c1-300
LM_DEBUG_MEM=1 timeout 1200 python bench.py --workload c3 --steps 2 --warmup 2 --no-cpu-baseline --ab-steps 2 --ab "LM_PA_FILTER_ROLL=0|LM_WFA_L2_MARGIN=4|LM_WFA_L2_MARGIN=8" > gpurun_out/r05_c3_ab2.json 2> gpurun_out/r05_c3_ab2.err; echo "c3 rc=$?"; grep -E "A/B|lane slabs" gpurun_out/r05_c3_ab2.err | cut -c1-250 | head -8
LM_TWO_LANES=0 LM_DEBUG_PA_CHAIN=1 timeout 300 python bench.py --workload c3mini --steps 1 --warmup 1 --no-cpu-baseline --no-exclusive-step > gpurun_out/r05_c3mini_clocks.json 2> gpurun_out/r05_c3mini_clocks.err; echo "c3mini clocks rc=$?"
LM_TWO_LANES=0 LM_DEBUG_PA_CHAIN=1 timeout 400 python bench.py --workload c3 --genomes 20000 --queries 2000 --families 201 --steps 1 --warmup 1 --no-cpu-baseline --no-exclusive-step > gpurun_out/r05_c3s_clocks2.json 2> gpurun_out/r05_c3s_clocks2.err; echo "c3s clocks rc=$?"
grep "k_pa_chain:" gpurun_out/r05_c3mini_clocks.err | tail -3 | cut -c1-600
grep "k_pa_chain:" gpurun_out/r05_c3s_clocks2.err | tail -4 | cut -c1-600
cd /tmp; rm -rf /tmp/prof_sq
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/prof_sq -o sq -- python $GRAFT_REPO_ROOT/bench.py --workload c2 --steps 1 --warmup 0 --no-cpu-baseline --no-exclusive-step > /tmp/prof_sq.log 2>&1; echo "c2 sq rc=$?"
cd $GRAFT_REPO_ROOT
H=$(python -c "import bench; print(bench.source_hash())")
python tools/summarize_rocprof.py /tmp/prof_sq gpurun_out/r05_c2_pmc_sq_interim.json $H "interim SQ pass (second GPU call)" | grep -E "^k_(wfa|pa_|extend|lookup|chain)" | cut -c1-250
python - <<'PY'
import json
for f in ("r05_c3_ab2", "r05_c3mini_clocks", "r05_c3s_clocks2"):
    try:
        d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], "first", d.get("first_step_ms"), d.get("warmup_step_ms"), d.get("step_ms"), "rows", d["rows"])
        print("   ab", d.get("ab"))
        print("   stage_ms", d["stage_ms"])
        rp = d["roofline_pipeline"]
        print("   kernel ms/step", rp["kernel_ms_per_step"], "exclusive", rp["exclusive_kernel_ms_per_step"])
        for k in d["kernels"][:12]:
            print("    %-22s launches %6d avg %9.3f ms/step %9.1f excl/step %s" % (k["name"], k["launches"], k["avg_ms"], k["ms_per_step"], k["exclusive_ms_per_step"]))
    except Exception as e:
        print(f, "no line:", e)
PY
