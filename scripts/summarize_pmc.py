#!/usr/bin/env python3
"""summarize_pmc.py <tag> <command> -- condense the rocprofv3 passes of scripts/profile_bench.sh for the dr_ kernel:
average duration from the --stats pass, per-launch averages of every counter, HBM traffic per launch =
FETCH_SIZE x 1 KiB x 2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE x 1 KiB.  Pure CSV parsing (no GPU)."""
import csv
import glob
import json
import os
import sys

tag, command = sys.argv[1], sys.argv[2]
root = os.path.join("gpurun_out", "prof_" + tag)
csv.field_size_limit(1 << 30)


def one(pattern):
    f = glob.glob(os.path.join(root, pattern))
    return f[0] if f else None


out = {"command": "rocprofv3 --kernel-trace --stats -- " + command}
bench = None
for line in open(os.path.join(root, "trace.log"), errors="replace"):
    if line.startswith("{") and '"metric"' in line:
        bench = json.loads(line)
if bench:
    out["generator_options"] = bench["config"]["generator_options"]
    out["workload"] = bench["config"]["workload"]
    out["bench_avg_launch_ms_same_run_hip_events"] = bench["roofline"]["avg_launch_ms"]
    out["algorithmic_bytes_per_launch"] = bench["roofline"]["algorithmic_bytes_per_launch"]
stats = one("trace/*/*kernel_stats.csv")
rows = list(csv.DictReader(open(stats)))
with open(os.path.join("gpurun_out", tag + "_kernel_stats.csv"), "w") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    for r in rows:
        r = dict(r)
        r["Name"] = r["Name"][:80]
        w.writerow(r)
for r in rows:
    if r["Name"].startswith("dr_"):
        out["kernel"] = r["Name"]
        out["calls"] = int(r["Calls"])
        out["avg_ns"] = float(r["AverageNs"])
        out["min_ns"] = float(r["MinNs"])
        out["max_ns"] = float(r["MaxNs"])
for sub in ("fetch", "write", "tcc", "sq"):
    f = one(sub + "/*/*counter_collection.csv")
    if not f:
        continue
    acc = {}
    for r in csv.DictReader(open(f)):
        if not r["Kernel_Name"].startswith("dr_"):
            continue
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        out.setdefault("vgpr", r["VGPR_Count"]); out.setdefault("lds", r["LDS_Block_Size"])
        out.setdefault("grid", r["Grid_Size"]); out.setdefault("wg", r["Workgroup_Size"])
    for k, v in acc.items():
        out[k] = sum(v) / len(v)
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    out["fetch_bytes_corrected_x2"] = out["FETCH_SIZE"] * 1024 * 2
    out["write_bytes"] = out["WRITE_SIZE"] * 1024
    out["traffic_bytes_per_launch"] = out["fetch_bytes_corrected_x2"] + out["write_bytes"]
    if "algorithmic_bytes_per_launch" in out:
        out["traffic_over_algorithmic"] = out["traffic_bytes_per_launch"] / out["algorithmic_bytes_per_launch"]
if "TCC_HIT_sum" in out:
    out["l2_hit_rate"] = out["TCC_HIT_sum"] / (out["TCC_HIT_sum"] + out["TCC_MISS_sum"])
if "SQ_WAIT_ANY" in out:
    out["wait_any_frac"] = out["SQ_WAIT_ANY"] / out["SQ_WAVE_CYCLES"]
    out["lds_conflict_frac"] = out["SQ_LDS_BANK_CONFLICT"] / max(out["SQ_LDS_IDX_ACTIVE"], 1)
json.dump(out, open(os.path.join("gpurun_out", tag + "_counters.json"), "w"), indent=1)
print(json.dumps({k: out[k] for k in out if k in ("kernel", "avg_ns", "bench_avg_launch_ms_same_run_hip_events", "traffic_over_algorithmic", "l2_hit_rate", "wait_any_frac", "vgpr")}))
