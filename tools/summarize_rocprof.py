#!/usr/bin/env python3
"""Condense rocprofv3 CSV output into small per-kernel summaries that are committed under profiles/.
  kernel stats : <prefix>_kernel_stats.csv          -> top kernels by total time (name shortened)
  PMC passes   : <prefix>_counter_collection.csv    -> mean counter value per kernel per dispatch
usage: summarize_rocprof.py <dir> <out.json> [source_hash] [command]"""
import csv
import glob
import json
import os
import re
import sys


def short(name):
    # k_wfa_lean2<NC, cell type, WIN> / k_wfa_mw2<NCW, WIN> (the restructured forward passes): the names of the kernels they replace
    m = re.search(r"lm::k_wfa_lean2<(\d+), *[a-z_ ]+, *(true|false|\(bool\)[01]|[01])>", name)
    if m:
        win = m.group(2).replace("(bool)", "") in ("true", "1")
        if win:
            return "k_wfa_win" + {"1": "64", "2": "128", "4": "256", "8": "512", "16": "1024"}.get(m.group(1), m.group(1))
        return "k_wfa_lean" + {"1": "64", "2": "", "4": "256", "8": "512", "16": "1024"}.get(m.group(1), m.group(1))
    m = re.search(r"lm::k_wfa_mw2<(\d+), *(true|false|\(bool\)[01]|[01])>", name)
    if m:
        win = m.group(2).replace("(bool)", "") in ("true", "1")
        return ("k_wfa_mww" if win else "k_wfa_mw") + {"2": "512", "4": "1024"}.get(m.group(1), m.group(1))
    m = re.search(r"lm::(k_[a-z0-9_]+)(?:<(\d+)(?:, *(true|false|\(bool\)[01]|[01]))?(?:, *[a-z_ ]+)?>)?", name)  # (<NC, WIN, cell type>)
    if m:
        if m.group(1) == "k_wfa_lean" and m.group(2):  # the names bench.py reports: k_wfa_lean<NC, WIN> -> diagonals of the ring
            win = (m.group(3) or "").replace("(bool)", "") in ("true", "1")  # (128 = plain); WIN = sliding sequence windows
            if win:
                return "k_wfa_win" + {"1": "64", "2": "128", "4": "256", "8": "512", "16": "1024"}.get(m.group(2), m.group(2))
            return "k_wfa_lean" + {"1": "64", "2": "", "4": "256", "8": "512", "16": "1024"}.get(m.group(2), m.group(2))
        if m.group(1) == "k_wfa_mw" and m.group(2):  # k_wfa_mw<NCW, WIN>: four wavefronts per alignment, 256 * NCW diagonals
            win = (m.group(3) or "").replace("(bool)", "") in ("true", "1")
            return ("k_wfa_mww" if win else "k_wfa_mw") + {"2": "512", "4": "1024"}.get(m.group(2), m.group(2))
        if m.group(1) == "k_pa_chain_wave":
            return "k_pa_chain"
        return m.group(1)
    m = re.search(r"(radix_sort_\w+|segmented_radix_sort\w*|scan_impl|reduce_by_key\w*|merge_sort\w*|init_lookback\w*)", name)
    if m:
        return "rocprim:" + m.group(1)
    return name[:60]


def main():
    d, out = sys.argv[1], sys.argv[2]
    res = {}
    if len(sys.argv) > 3:
        res["source_hash"] = sys.argv[3]  # bench.py source_hash(): the kernels these numbers were taken on
    if len(sys.argv) > 4:
        res["command"] = sys.argv[4]
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        agg = {}
        for r in rows:
            k = short(r["Name"])
            a = agg.setdefault(k, dict(calls=0, total_ns=0))
            a["calls"] += int(r["Calls"])
            a["total_ns"] += int(r["TotalDurationNs"])
        tot = sum(a["total_ns"] for a in agg.values()) or 1
        res["kernel_stats"] = [dict(name=k, calls=a["calls"], total_ms=round(a["total_ns"] / 1e6, 3),
                                    avg_ms=round(a["total_ns"] / a["calls"] / 1e6, 4), pct=round(100 * a["total_ns"] / tot, 2))
                               for k, a in sorted(agg.items(), key=lambda x: -x[1]["total_ns"])]
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = {}
        for r in csv.DictReader(open(f)):
            k = short(r.get("Kernel_Name", ""))
            c = r.get("Counter_Name", "")
            v = float(r.get("Counter_Value", 0) or 0)
            a = agg.setdefault((k, c), [0, 0.0])
            a[0] += 1
            a[1] += v
        pm = res.setdefault("pmc", {})
        for (k, c), (n, s) in agg.items():
            pm.setdefault(k, {})[c] = dict(dispatches=n, mean=s / n, total=s)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res.get("kernel_stats", [])[:8], indent=0))
    for k, v in res.get("pmc", {}).items():
        if k.startswith("k_"):
            print(k, {c: round(x["mean"], 1) for c, x in v.items()})


if __name__ == "__main__":
    main()
