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
import hashlib
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha16():
    """Fingerprint of the kernel sources a profile was taken from (bench.py quotes a committed profile only while it matches)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "rnnt-speech-recognition_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def short(name):
    name = re.sub(r"\(.*$", "", name)          # drop the argument list
    name = name.replace("void ", "")
    return name[:120]


HEADLINE = {  # names the bench's roofline object refers to
    "grad_pass": ("cell_tile_kernel", "true"),
    "lsm_pass": ("cell_tile_kernel", "false"),
    "sweeps": ("sweep_kernel",),
    "sweeps_log": ("sweep_ld_kernel",),
    "redo": ("lin_redo_kernel",),
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
    json.dump({"source": "rocprofv3 --kernel-trace --stats", "csrc_sha16": csrc_sha16(), "headline_kernels": head, "kernels": kernels[:40]},
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
    json.dump({"source": "rocprofv3 --pmc (one counter group per pass)", "csrc_sha16": csrc_sha16(), "kernels": out}, open(out_json, "w"), indent=1)
    for kern, c in out.items():
        if any(s in kern for s in ("rnnt", "fill")):
            print(kern[:80], {k: round(v["avg"]) for k, v in c.items()})


def latest(pmc_json, out_json, cells_bytes):
    """HBM bytes per launch of the op-level kernels from a FETCH_SIZE / WRITE_SIZE pmc summary (KiB; FETCH doubled: on gfx950
    FETCH_SIZE reports half of the bytes of wide streaming reads, /opt/skills/guides/MI355X_MICROARCH.md)."""
    d = json.load(open(pmc_json))
    roles = {"grad": (r"cell_tile_kernel<\d+, true",), "lsm": (r"cell_tile_kernel<\d+, false",), "sweep": (r"sweep_kernel",),
             "redo": (r"lin_redo_kernel",)}
    out = {"source": f"{os.path.basename(pmc_json)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, KiB; FETCH doubled per "
                     "the gfx950 note of MI355X_MICROARCH.md)", "csrc_sha16": d.get("csrc_sha16")}
    step = 0.0
    for role, needles in roles.items():
        for kern, c in d["kernels"].items():
            if all(re.search(n, kern) for n in needles) and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                b = (2.0 * c["FETCH_SIZE"]["avg"] + c["WRITE_SIZE"]["avg"]) * 1024.0
                out[f"{role}_kernel_hbm_bytes_per_launch"] = b
                out[f"{role}_kernel"] = kern
                step += b
                break
    out["step_hbm_bytes"] = step
    out["algorithmic_bytes_per_step"] = float(cells_bytes)
    out["step_traffic_over_algorithmic"] = step / float(cells_bytes)
    json.dump(out, open(out_json, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "latest":
        latest(sys.argv[2], sys.argv[3], float(sys.argv[4]) if len(sys.argv) > 4 else 645120000.0)
    elif mode == "stats":
        stats(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else None)
    else:
        pmc(sys.argv[2], sys.argv[3])
