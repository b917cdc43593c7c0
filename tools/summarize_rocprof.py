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


_DIAG = {"1": "64", "2": "128", "4": "256", "8": "512", "16": "1024"}


def _targs(name, kernel):
    """template arguments of `lm::<kernel><...>` in a demangled kernel name (None when it is no such instantiation): whatever
    their number - k_wfa_lean2 has five (NC, cell type, WIN, shrink margin, wavefronts per SIMD), the regular expression of round
    5 knew three and lumped every instantiation under one name"""
    m = re.search(r"lm::%s<([^<>()]*(?:\(bool\)[01][^<>()]*)*)>" % re.escape(kernel), name)
    if not m:
        return None
    return [a.strip() for a in m.group(1).split(",")]


def _flag(a):
    return a.replace("(bool)", "").strip() in ("true", "1")


def short(name):
    """the name bench.py (the library's profile) reports for a kernel: ONE summary name per bench name"""
    a = _targs(name, "k_wfa_lean2")  # <NC, cell type, WIN, ...>: k_wfa_lean (128 diagonals) / k_wfa_lean<diagonals> / k_wfa_win<diagonals>
    if a is not None and len(a) >= 3:
        if _flag(a[2]):
            return "k_wfa_win" + _DIAG.get(a[0], a[0])
        return "k_wfa_lean" + ("" if a[0] == "2" else _DIAG.get(a[0], a[0]))
    a = _targs(name, "k_wfa_mw2")  # <NCW, WIN, ...>: four wavefronts per alignment, 256 * NCW diagonals
    if a is not None and len(a) >= 2:
        return ("k_wfa_mww" if _flag(a[1]) else "k_wfa_mw") + {"2": "512", "4": "1024"}.get(a[0], a[0])
    if re.search(r"lm::k_wfa_(lean2|mw2)\b", name):
        return name[:60]  # an instantiation this table does not know: never under a neighbour's name
    m = re.search(r"lm::(k_[a-z0-9_]+)", name)
    if m:
        if m.group(1) == "k_pa_chain_wave":
            return "k_pa_chain"
        return m.group(1)
    m = re.search(r"(radix_sort_\w+|segmented_radix_sort\w*|scan_impl|reduce_by_key\w*|merge_sort\w*|init_lookback\w*)", name)
    if m:
        return "rocprim:" + m.group(1)
    return name[:60]


MARK = "k_profile_mark"  # lm_profile_mark(): bench.py launches one in front of and one behind its timed steps


def _order(r):
    """position of a dispatch in time: its start time stamp when the file has one, else the dispatch id"""
    for f in ("Start_Timestamp", "Dispatch_Id"):
        if r.get(f) not in (None, ""):
            return int(r[f])
    return 0


def window(rows, name_field):
    """the rows between the first two marker dispatches (the timed - warm - steps of bench.py), or all rows when the run has
    no markers.  Returns (rows, note)."""
    marks = sorted({_order(r) for r in rows if MARK in r.get(name_field, "")})
    if len(marks) < 2:
        return [r for r in rows if MARK not in r.get(name_field, "")], None
    lo, hi = marks[0], marks[1]
    return [r for r in rows if lo < _order(r) < hi and MARK not in r.get(name_field, "")], \
        "dispatches between the two lm::k_profile_mark kernels only = the timed steps, after the warm-up"


def main():
    d, out = sys.argv[1], sys.argv[2]
    res = {}
    if len(sys.argv) > 3:
        res["source_hash"] = sys.argv[3]  # bench.py source_hash(): the kernels these numbers were taken on
    if len(sys.argv) > 4:
        res["command"] = sys.argv[4]
        m = re.search(r"--steps[ =](\d+)", sys.argv[4])
        if m:
            res["window_steps"] = int(m.group(1))  # (meaningful with a window: the steps between the markers)
    agg = None
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):  # per dispatch: can be cut at the markers
        rows, note = window(list(csv.DictReader(open(f))), "Kernel_Name")
        if note is None:
            continue
        res["window"] = note
        agg = agg or {}
        for r in rows:
            a = agg.setdefault(short(r["Kernel_Name"]), dict(calls=0, total_ns=0))
            a["calls"] += 1
            a["total_ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if agg is None:
        for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
            agg = agg or {}
            for r in csv.DictReader(open(f)):
                if MARK in r["Name"]:
                    continue
                a = agg.setdefault(short(r["Name"]), dict(calls=0, total_ns=0))
                a["calls"] += int(r["Calls"])
                a["total_ns"] += int(r["TotalDurationNs"])
    if agg:
        tot = sum(a["total_ns"] for a in agg.values()) or 1
        res["kernel_stats"] = [dict(name=k, calls=a["calls"], total_ms=round(a["total_ns"] / 1e6, 3),
                                    avg_ms=round(a["total_ns"] / a["calls"] / 1e6, 4), pct=round(100 * a["total_ns"] / tot, 2))
                               for k, a in sorted(agg.items(), key=lambda x: -x[1]["total_ns"])]
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows, note = window(list(csv.DictReader(open(f))), "Kernel_Name")
        if note:
            res["window"] = note
        agg = {}
        for r in rows:
            k = short(r.get("Kernel_Name", ""))
            c = r.get("Counter_Name", "")
            v = float(r.get("Counter_Value", 0) or 0)
            a = agg.setdefault((k, c), [0, 0.0])
            a[0] += 1
            a[1] += v
        pm = res.setdefault("pmc", {})
        for (k, c), (n, s) in agg.items():
            pm.setdefault(k, {})[c] = dict(dispatches=n, mean=s / n, total=s)
    if "window" not in res:
        res.pop("window_steps", None)  # the whole run was summed: not a per-step window
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res.get("kernel_stats", [])[:8], indent=0))
    for k, v in res.get("pmc", {}).items():
        if k.startswith("k_"):
            print(k, {c: round(x["mean"], 1) for c, x in v.items()})


if __name__ == "__main__":
    main()
