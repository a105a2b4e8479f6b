#!/usr/bin/env python
"""Summarise rocprofv3 output directories into the small JSON / CSV files committed under profiles/.

  summarize_trace.py stats  <rocprof dir> <out.json> [<out.csv>]   kernel-trace --stats  -> per-kernel calls / avg / min / max
  summarize_trace.py pmc    <rocprof dir> <out.json>               --pmc pass(es)        -> per-kernel per-counter averages
Kernel names are shortened to the part a reader needs (namespace + template arguments kept)."""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)          # drop the argument list
    name = name.replace("void ", "")
    return name[:120]


HEADLINE = {  # names the bench's roofline object refers to
    "grad_pass": ("cell_tile_kernel", "true"),
    "lsm_pass": ("cell_tile_kernel", "false"),
    "sweeps": ("sweep_ld_kernel",),
    "fill": ("fillBuffer",),
    "step": ("rnnt_step_kernel",),
}


def stats(d, out_json, out_csv=None):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append(r)
    if not rows:  # fall back to the per-dispatch trace
        agg = collections.defaultdict(list)
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                agg[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for k, v in agg.items():
            rows.append({"Name": k, "Calls": len(v), "TotalDurationNs": sum(v), "AverageNs": sum(v) / len(v),
                         "MinNs": min(v), "MaxNs": max(v)})
    kernels = []
    for r in rows:
        kernels.append({"kernel": short(r["Name"]), "calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) * 1e-6,
                        "min_ms": float(r["MinNs"]) * 1e-6, "max_ms": float(r["MaxNs"]) * 1e-6,
                        "total_ms": float(r["TotalDurationNs"]) * 1e-6})
    kernels.sort(key=lambda k: -k["total_ms"])
    head = {}
    for key, needles in HEADLINE.items():
        for k in kernels:
            if all(n in k["kernel"] for n in needles):
                head[key] = {"kernel": k["kernel"], "calls": k["calls"], "avg_ms": k["avg_ms"], "min_ms": k["min_ms"],
                             "max_ms": k["max_ms"]}
                break
    json.dump({"source": "rocprofv3 --kernel-trace --stats", "headline_kernels": head, "kernels": kernels[:40]},
              open(out_json, "w"), indent=1)
    if out_csv:
        with open(out_csv, "w") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "calls", "avg_us", "min_us", "max_us", "total_ms"])
            for k in kernels[:40]:
                w.writerow([k["kernel"], k["calls"], f"{k['avg_ms'] * 1e3:.2f}", f"{k['min_ms'] * 1e3:.2f}",
                            f"{k['max_ms'] * 1e3:.2f}", f"{k['total_ms']:.3f}"])
    for k in kernels[:14]:
        print(f"{k['avg_ms'] * 1e3:10.1f} us x{k['calls']:5d}  {k['kernel'][:90]}")


def pmc(d, out_json):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = (short(r["Kernel_Name"]), r["Counter_Name"])
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
    out = collections.defaultdict(dict)
    for (kern, ctr), (n, tot) in sorted(agg.items()):
        out[kern][ctr] = {"launches": n, "avg": tot / n}
    json.dump({"source": "rocprofv3 --pmc (one counter group per pass)", "kernels": out}, open(out_json, "w"), indent=1)
    for kern, c in out.items():
        if any(s in kern for s in ("rnnt", "fill")):
            print(kern[:80], {k: round(v["avg"]) for k, v in c.items()})


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "stats":
        stats(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else None)
    else:
        pmc(sys.argv[2], sys.argv[3])
