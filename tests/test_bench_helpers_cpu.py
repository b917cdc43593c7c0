"""CPU tests of bench.py's own checking / bookkeeping helpers (no GPU, no library): the row-for-row comparison that guards the
bench's parity claim, and the kernel-name mapping of the rocprofv3 summaries the roofline block reads."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _rows(n):
    import bench
    dt = np.dtype([(f, "<f8" if f in ("qcov_genome", "qcov_hsp", "pident") else "<i8") for f in bench.ROW_CHECK_FIELDS] +
                  [("evalue", "<f8"), ("hits", "<i8"), ("query", "<i8")])
    a = np.zeros(n, dtype=dt)
    for i in range(n):
        for j, f in enumerate(bench.ROW_CHECK_FIELDS):
            a[f][i] = (i * 7 + j) % 50
        a["evalue"][i] = 1e-30 * (i + 1)
        a["hits"][i] = 3
        a["query"][i] = i // 4
    return a


def _oracle_of(a):
    import bench
    out = {}
    for r in a:
        out.setdefault(int(r["query"]), []).append(tuple(r[f].item() for f in bench.ROW_CHECK_FIELDS) + (float(r["evalue"]), int(r["hits"])))
    return out


def test_rows_equal_oracle_accepts_equal_rows_and_names_the_first_difference():
    import bench
    a = _rows(12)
    exp = _oracle_of(a)
    ok, n, diff = bench.rows_equal_oracle(a, exp)
    assert ok and n == 12 and diff is None
    # e-value within 1e-9 relative passes, beyond fails
    b = a.copy()
    b["evalue"][5] *= 1 + 1e-12
    assert bench.rows_equal_oracle(b, exp)[0]
    b["evalue"][5] *= 1 + 1e-6
    ok, n, diff = bench.rows_equal_oracle(b, exp)
    assert not ok and "evalue" in diff and "query 1 row 1" in diff
    # an integer column, the hits column, a missing row, a query without rows on one side
    b = a.copy()
    b["tend"][9] += 1
    ok, n, diff = bench.rows_equal_oracle(b, exp)
    assert not ok and "tend" in diff and n == 9
    b = a.copy()
    b["hits"][0] = 4
    assert "hits" in bench.rows_equal_oracle(b, exp)[2]
    assert "11 HIP rows" not in str(bench.rows_equal_oracle(a[:11], exp)[2]) and not bench.rows_equal_oracle(a[:11], exp)[0]
    exp2 = dict(exp)
    exp2[7] = []                      # the oracle found nothing for query 7, neither did the HIP path
    assert bench.rows_equal_oracle(a, exp2)[0]
    exp2[7] = [exp[0][0]]             # ... but now it did
    assert not bench.rows_equal_oracle(a, exp2)[0]


def test_rows_equal_oracle_translates_genome_keys_between_a_sample_index_and_the_full_index(tmp_path):
    """the full-index check: the oracle's rows carry the genome keys of the SAMPLE index (j-th genome = j), the HIP rows those
    of the full bench index; info.toml's database size is set to the full index's"""
    import bench
    a = _rows(12)
    exp = _oracle_of(a)                      # keys as the sample index numbers them
    keymap = {k: bench.genome_key(1001 * k) for k in range(50)}
    b = a.copy()
    b["batch_genome"] = [keymap[int(k)] for k in a["batch_genome"]]
    assert not bench.rows_equal_oracle(b, exp)[0]
    ok, n, diff = bench.rows_equal_oracle(b, exp, keymap)
    assert ok and n == 12 and diff is None
    assert bench.genome_key(4999) == 4999 and bench.genome_key(5000) == 1 << 17 and bench.genome_key(99_999) == (19 << 17) | 4999
    t = tmp_path / "info.toml"
    t.write_text("k = 31\ninput-bases = 1234\nmasks = 20000\n")
    bench.set_input_bases(str(t), 200_000_000_000)
    assert t.read_text() == "k = 31\ninput-bases = 200000000000\nmasks = 20000\n"


def test_rocprof_kernel_names_match_the_names_bench_reports():
    import summarize_rocprof as S
    # k_wfa_lean2 / k_wfa_mw2 keep the profile names of the kernels they replaced in round 5; any number of trailing template
    # arguments (round 5's pattern knew three, the kernel has five: every instantiation landed under ONE name)
    assert S.short("void lm::k_wfa_lean2<2, short, false>(x)") == "k_wfa_lean"
    assert S.short("void lm::k_wfa_lean2<2, short, false, 12, 8>(lm::WfaIn const*, long, int const*)") == "k_wfa_lean"
    assert S.short("void lm::k_wfa_lean2<4, short, false, 12, 1>(lm::WfaIn const*, long)") == "k_wfa_lean256"
    assert S.short("void lm::k_wfa_lean2<4, int, true, 12, 1>(lm::WfaIn const*, long)") == "k_wfa_win256"
    assert S.short("void lm::k_wfa_lean2<4, int, (bool)1>(x)") == "k_wfa_win256"
    assert S.short("void lm::k_wfa_lean2<1, int, false, 12, 1>(x)") == "k_wfa_lean64"
    assert S.short("void lm::k_wfa_lean2<16, int, (bool)0, 12, 1>(x)") == "k_wfa_lean1024"
    assert S.short("void lm::k_wfa_mw2<4, true>(x)") == "k_wfa_mww1024"
    assert S.short("void lm::k_wfa_mw2<2, false>(lm::WfaIn const*, long)") == "k_wfa_mw512"
    assert S.short("void lm::k_pa_chain_wave<true>(unsigned long const*, ...)") == "k_pa_chain"
    assert S.short("void lm::k_pa_filter<true>(lm::DevIndexView, lm::Task const*, long)") == "k_pa_filter"
    assert S.short("lm::k_pa_search(lm::DevIndexView, ...)") == "k_pa_search"
    assert S.short("void rocprim::detail::radix_sort_onesweep_kernel<...>").startswith("rocprim:")


def test_every_kernel_of_a_committed_trace_gets_one_summary_name_per_bench_name():
    """the kernel names rocprofv3 printed for the round-5 C3 run: each WFA instantiation must land under the name the library's
    profile (bench.py) reports for it, and no two instantiations under one name"""
    import csv
    import summarize_rocprof as S
    names = [r["Name"] for r in csv.DictReader(open(os.path.join(ROOT, "profiles", "r05_c3_kernel_stats.csv")))]
    wfa = [n for n in names if "k_wfa_" in n]
    got = [S.short(n) for n in wfa]
    assert len(wfa) >= 6 and len(set(got)) == len(wfa), got
    bench_names = {"k_wfa_lean64", "k_wfa_lean", "k_wfa_lean256", "k_wfa_lean512", "k_wfa_lean1024", "k_wfa_win64", "k_wfa_win128",
                   "k_wfa_win256", "k_wfa_win512", "k_wfa_win1024", "k_wfa_mw512", "k_wfa_mw1024", "k_wfa_mww512", "k_wfa_mww1024", "k_wfa_wave"}
    assert set(got) <= bench_names, set(got) - bench_names
    assert {"k_wfa_lean", "k_wfa_lean256", "k_wfa_win256"} <= set(got)


def test_a_pass_is_cut_at_the_step_markers(tmp_path):
    """counters / durations of the timed (warm) steps only: the dispatches between the two lm::k_profile_mark kernels"""
    import csv
    import json
    import subprocess
    d = tmp_path / "out"
    d.mkdir()
    rows = []
    seq = ["lm::k_mask(a)"] * 3 + ["lm::k_profile_mark(int)"] + ["lm::k_mask(a)", "void lm::k_wfa_lean2<2, short, false, 12, 8>(x)"] * 2 + \
          ["lm::k_profile_mark(int)"] + ["lm::k_mask(a)"] * 5
    for i, n in enumerate(seq):
        rows.append(dict(Dispatch_Id=i + 1, Kernel_Name=n, Counter_Name="SQ_INSTS_SALU", Counter_Value=10 + i))
    with open(d / "x_counter_collection.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0]))
        w.writeheader()
        w.writerows(rows)
    with open(d / "x_kernel_trace.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["Dispatch_Id", "Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        w.writeheader()
        for i, n in enumerate(seq):
            w.writerow(dict(Dispatch_Id=i + 1, Kernel_Name=n, Start_Timestamp=1000 * i, End_Timestamp=1000 * i + 100 + i))
    out = tmp_path / "s.json"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "summarize_rocprof.py"), str(d), str(out), "hash", "python bench.py --steps 2 --warmup 1"],
                          stdout=subprocess.DEVNULL)
    doc = json.load(open(out))
    assert "between the two" in doc["window"] and doc["window_steps"] == 2
    assert doc["pmc"]["k_mask"]["SQ_INSTS_SALU"]["dispatches"] == 2 and doc["pmc"]["k_wfa_lean"]["SQ_INSTS_SALU"]["dispatches"] == 2
    assert doc["pmc"]["k_mask"]["SQ_INSTS_SALU"]["total"] == (10 + 4) + (10 + 6)
    ks = {k["name"]: k for k in doc["kernel_stats"]}
    assert ks["k_mask"]["calls"] == 2 and ks["k_wfa_lean"]["calls"] == 2 and "k_profile_mark" not in ks
    # a run without markers is summed whole and carries no window
    for f in (d / "x_counter_collection.csv", d / "x_kernel_trace.csv"):
        f.write_text(f.read_text().replace("lm::k_profile_mark(int)", "lm::k_other(int)"))
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "summarize_rocprof.py"), str(d), str(out), "hash", "python bench.py --steps 2"],
                          stdout=subprocess.DEVNULL)
    doc = json.load(open(out))
    assert "window" not in doc and "window_steps" not in doc and doc["pmc"]["k_mask"]["SQ_INSTS_SALU"]["dispatches"] == 10


def test_workloads_name_every_baseline_config_and_the_shards_fit():
    """BASELINE.json configs[1..4] = c2, c3, c4, c5; a C4 / C5 shard must fit 288 GB at the measured 9.45 B per seed"""
    import bench
    wl = bench.WORKLOADS
    assert wl["c2"]["genomes"] == 10_000 and wl["c3"]["genomes"] == 100_000 and wl["c3"]["queries"] == 10_000
    assert wl["c4"]["genomes"] == 1_000_000 and wl["c4"]["shards"] == 4 and wl["c4"]["qlen"] == (50_000, 200_000)
    assert wl["c5"]["genomes"] == 1_900_000 and wl["c5"]["shards"] == 8 and wl["c5"]["kind"] == "mixed"
    for w in ("c4", "c5"):
        per_shard = wl[w]["genomes"] / wl[w]["shards"]
        seeds = per_shard * 2 * (20_000 + wl[w]["genome_len"] / 77)      # captures + desert seeds, x2 reversed twins
        resident = seeds * 9.45 + per_shard * wl[w]["genome_len"] / 4
        assert resident < 0.7 * 288e9, (w, resident)                      # leaves >= 30 % of the HBM for scratch
        assert wl[w]["families"] % 2 == 1                                 # family members spread over 2 / 4 / 8 shards


def test_counter_passes_are_spread_over_this_runs_launches_per_step(tmp_path, monkeypatch):
    """roofline.traffic / instruction_issue: the committed passes time ONE step; a run whose step is cut into another number of
    launches (batch parts halved under memory pressure stay halved) must use the pass's per-step TOTAL over its own launches
    per step, not the pass's per-launch mean"""
    import json
    import bench
    prof = tmp_path / "profiles"
    prof.mkdir()
    h = bench.source_hash()
    mk = lambda ctr, n, mean: {"source_hash": h, "pmc": {"k_wfa_lean": {ctr: {"dispatches": n, "mean": mean, "total": n * mean}}}}
    (prof / ("%s_c3_pmc_fetch.json" % bench.PROFILE_ROUND)).write_text(json.dumps(mk("FETCH_SIZE", 100, 2000.0)))
    (prof / ("%s_c3_pmc_write.json" % bench.PROFILE_ROUND)).write_text(json.dumps(mk("WRITE_SIZE", 100, 1000.0)))
    sq = {"source_hash": h, "pmc": {"k_wfa_lean": {c: {"dispatches": 100, "mean": 5e6, "total": 5e8} for c in
                                                     ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS")}}}
    (prof / ("%s_c3_pmc_sq.json" % bench.PROFILE_ROUND)).write_text(json.dumps(sq))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "source_hash", lambda: h)
    per_launch, _ = bench.pmc_traffic("k_wfa_lean", "c3")
    per_step, note = bench.pmc_traffic("k_wfa_lean", "c3", per_step=True)
    assert per_launch == 3000 * 1024 and per_step == 100 * 3000 * 1024 and "launches per step" in note
    assert bench.pmc_issue("k_wfa_lean", "c3")["sq_insts_valu_per_launch"] == 5000000
    assert bench.pmc_issue("k_wfa_lean", "c3", per_step_launches=400)["sq_insts_valu_per_launch"] == 1250000
    # exact names only: a kernel the pass does not list gets nothing - never the counters of a neighbour whose name it starts
    assert bench.pmc_traffic("k_wfa_lean256", "c3", per_step=True)[0] is None and bench.pmc_issue("k_wfa_lean256", "c3") is None
    assert bench.pmc_traffic("k_wfa", "c3")[0] is None
    # a pass cut at the step markers over two steps: totals are per window, the bench wants them per step
    sq["window_steps"] = 2
    (prof / ("%s_c3_pmc_sq.json" % bench.PROFILE_ROUND)).write_text(json.dumps(sq))
    assert bench.pmc_issue("k_wfa_lean", "c3", per_step_launches=400)["sq_insts_valu_per_launch"] == 625000
    monkeypatch.setattr(bench, "source_hash", lambda: "other")   # passes of other sources are refused
    assert bench.pmc_traffic("k_wfa_lean", "c3", per_step=True)[0] is None and bench.pmc_issue("k_wfa_lean", "c3") is None


def test_source_hash_ignores_comments_and_whitespace_but_not_code():
    """bench.source_hash ties the committed counter passes to the library's CODE: comments and layout may change, a token may not"""
    import bench
    a = 'int f(int x) { // why\n    return x /* inline */ + 1; }\nconst char *s = "// not a comment";\n'
    b = 'int f(int x) {\n  return x + 1;   }  /* new\n comment */ const char *s = "// not a comment";'
    c = 'int f(int x) { return x + 2; }\nconst char *s = "// not a comment";'
    assert bench._code_only(a) == bench._code_only(b) != bench._code_only(c)
    assert '"// not a comment"' in bench._code_only(a)
    assert len(bench.source_hash()) == 16 and bench.source_hash() != bench.source_hash_raw()
