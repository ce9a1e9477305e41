#!/usr/bin/env python3
"""Turns gpurun_out/<tag>/ (written by scripts/profile_bench.sh on the GPU box) into the committed
profiles/<tag>_kernel_stats.csv + profiles/<tag>_summary.json.

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB and
come from separate --pmc passes; on gfx950 FETCH_SIZE reports half of the bytes a streaming read
fetches, so it is doubled; WRITE_SIZE is exact for 16-byte-per-lane streaming stores."""
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join("gpurun_out", tag)
os.makedirs("profiles", exist_ok=True)


def one(pattern):
    """the largest match (a profiled program that starts children leaves one file per process: the main process's is the big one)"""
    fs = sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getsize)
    return fs[-1] if fs else None


summary = {"tag": tag, "kernels": {}}
stats = one("trace/*/*_kernel_stats.csv")
if stats:
    shutil.copy(stats, f"profiles/{tag}_kernel_stats.csv")
    for r in csv.DictReader(open(stats)):
        summary["kernels"][r["Name"]] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
                                         "max_ns": float(r["MaxNs"]), "pct": float(r["Percentage"])}
trace = one("trace/*/*_kernel_trace.csv")
if trace:
    rows = list(csv.DictReader(open(trace)))
    res = {}
    for r in rows:
        res.setdefault(r["Kernel_Name"], {"vgpr": r.get("VGPR_Count"), "sgpr": r.get("SGPR_Count"), "lds": r.get("LDS_Block_Size"),
                                          "scratch": r.get("Scratch_Size"), "wg": r.get("Workgroup_Size"), "grid": r.get("Grid_Size")})
    for k, v in res.items():
        summary["kernels"].setdefault(k, {}).update(v)
def steps_of(log):
    """steps + warmup of the bench.py run a PMC pass profiled (its JSON line), so that counters can be given per STEP."""
    p = os.path.join(src, log)
    if os.path.exists(p):
        for line in open(p):
            if line.startswith("{") and '"metric"' in line:
                d = json.loads(line)
                return d["steps"] + d["warmup"]
    return None


# FETCH_SIZE x 2, WRITE_SIZE x 1 for every kernel: measured per access shape in profiles/r03_fetch_calibration.txt (2.000 / 1.000 for all
# sixteen load / store shapes the kernels use, scripts/fetch_calibration.sh)
for kind, key, log in (("fetch", "FETCH_SIZE", "bench_pmc_fetch.log"), ("write", "WRITE_SIZE", "bench_pmc_write.log")):
    f = one(f"pmc_{kind}/*/*_counter_collection.csv")
    if not f:
        continue
    shutil.copy(f, f"profiles/{tag}_pmc_{kind}.csv")
    nsteps = steps_of(log)
    acc = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == key:
            acc.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    for k, vals in acc.items():
        kib = sum(vals) / len(vals)
        d = summary["kernels"].setdefault(k, {})
        d[key + "_KiB_per_launch_raw"] = kib
        mul = 2 if key == "FETCH_SIZE" else 1
        d["hbm_read_bytes_per_launch" if key == "FETCH_SIZE" else "hbm_write_bytes_per_launch"] = kib * 1024 * mul
        if nsteps:      # per STEP of the workload (a step may launch a kernel several times, or several kernels): total over the run / steps of the run
            d["hbm_read_bytes_per_step" if key == "FETCH_SIZE" else "hbm_write_bytes_per_step"] = sum(vals) * 1024 * mul / nsteps
for k, d in summary["kernels"].items():
    if "hbm_read_bytes_per_launch" in d and "hbm_write_bytes_per_launch" in d:
        d["hbm_bytes_per_launch"] = d["hbm_read_bytes_per_launch"] + d["hbm_write_bytes_per_launch"]
    if "hbm_read_bytes_per_step" in d and "hbm_write_bytes_per_step" in d:
        d["hbm_bytes_per_step"] = d["hbm_read_bytes_per_step"] + d["hbm_write_bytes_per_step"]
for log in ("bench_trace.log",):
    p = os.path.join(src, log)
    if os.path.exists(p):
        for line in open(p):
            if line.startswith("{") and '"metric"' in line:
                summary["bench_line_under_profiler"] = json.loads(line)
json.dump(summary, open(f"profiles/{tag}_summary.json", "w"), indent=1)
print(json.dumps({k[:60]: {kk: vv for kk, vv in v.items() if kk in ("avg_ns", "hbm_bytes_per_launch", "vgpr", "calls")}
                  for k, v in summary["kernels"].items()}, indent=1))
