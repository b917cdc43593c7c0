#!/usr/bin/env python3
"""profiles/<round>_<workload>_bench.json is written by the first run of tools/profile.sh, BEFORE the FETCH/WRITE/SQ counter
passes of the same sources exist, so its roofline objects say "traffic: null".  This fills those fields in afterwards from
the committed passes with bench.py's own pmc_traffic / pmc_issue (same rules: the bench line, the passes and the tree must
carry the same source hash) and marks every filled object.  No GPU, no timing: durations and bytes stay as measured.

usage: python tools/refill_traffic.py c3 [c2 ...]"""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
argv, sys.argv = sys.argv, ["bench.py"]
spec.loader.exec_module(bench)
sys.argv = argv

NOTE = "filled after the run by tools/refill_traffic.py: the counter passes of the same sources were taken after this bench line"


def fill(obj, workload):
    if not isinstance(obj, dict) or "kernel" not in obj:
        return 0
    n = 0
    dur_ms, alg = obj.get("avg_launch_ms"), obj.get("algorithmic_bytes_per_launch")
    if obj.get("traffic") is None:
        tr, tn = bench.pmc_traffic(obj["kernel"], workload)
        if tr:
            obj["traffic"], obj["traffic_source"] = tr, tn
            obj["traffic_over_algorithmic"] = round(tr / alg, 2) if alg else None
            obj["traffic_GBs"] = round(tr / (dur_ms * 1e-3) / 1e9, 1) if dur_ms else None
            obj["traffic_filled"] = NOTE
            n += 1
    if obj.get("instruction_issue") is None:
        issue = bench.pmc_issue(obj["kernel"], workload)
        if issue and dur_ms:
            issue["issue_frac_valu"] = round(issue.get("sq_insts_valu_per_launch", 0) / (dur_ms * 1e-3) / bench.VALU_ISSUE_PEAK, 4)
            issue["issue_frac_salu"] = round(issue.get("sq_insts_salu_per_launch", 0) / (dur_ms * 1e-3) / bench.SALU_ISSUE_PEAK, 4)
            issue["issue_frac_valu_measured_mix"] = round(issue.get("sq_insts_valu_per_launch", 0) / (dur_ms * 1e-3) / bench.VALU_ISSUE_PEAK_MIX, 4)
            issue["issue_filled"] = NOTE
            obj["instruction_issue"] = issue
            n += 1
    iss = obj.get("instruction_issue")
    if isinstance(iss, dict) and "issue_frac_valu_measured_mix" not in iss and dur_ms:
        iss["issue_frac_valu_measured_mix"] = round(iss.get("sq_insts_valu_per_launch", 0) / (dur_ms * 1e-3) / bench.VALU_ISSUE_PEAK_MIX, 4)
        iss["issue_filled"] = NOTE
        n += 1
    for sib in obj.get("other_instantiations", []):
        n += fill(sib, workload)
    return n


def main():
    for wl in sys.argv[1:]:
        path = os.path.join(ROOT, "profiles", "%s_%s_bench.json" % (bench.PROFILE_ROUND, wl))
        doc = json.load(open(path))
        if doc.get("source_hash") != bench.source_hash():
            sys.exit("%s was measured on sources %s, the tree is %s: refused" % (path, doc.get("source_hash"), bench.source_hash()))
        n = sum(fill(doc.get(k), wl) for k in ("roofline", "roofline_seed_lookup"))
        json.dump(doc, open(path, "w"))
        open(path, "a").write("\n")
        r = doc["roofline"]
        print("%s: %d fields filled; %s traffic %s B/launch (%sx algorithmic), issue %s" % (
            path, n, r["kernel"], r.get("traffic"), r.get("traffic_over_algorithmic"), r.get("instruction_issue")))


if __name__ == "__main__":
    main()
