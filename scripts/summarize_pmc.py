#!/usr/bin/env python3
"""summarize_pmc.py <tag> <command> [steps] -- condense the rocprofv3 passes of scripts/profile_bench.sh for the dr_ kernel.

Durations come from the per-dispatch kernel trace of the --stats pass with the WARM-UP DISPATCHES EXCLUDED: only the last
steps * launches_per_step dispatches (bench.py's timed region) are averaged, so `timed_avg_ns` is comparable with the same
run's HIP-event figure (`bench_avg_launch_ms_same_run_hip_events`); min / median / max of those dispatches are recorded too,
next to the all-dispatch averages rocprofv3's own stats file reports (which mix in the cold warm-up launches).
Counters: per-launch averages over the timed dispatches of every --pmc pass found; HBM traffic per launch =
FETCH_SIZE x 1 KiB x 2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE x 1 KiB.  Pure CSV parsing (no GPU)."""
import csv
import glob
import json
import os
import statistics
import sys

tag, command = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else None
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
timed = None
if bench:
    out["generator_options"] = bench["config"]["generator_options"]
    out["workload"] = bench["config"]["workload"]
    out["bench_avg_launch_ms_same_run_hip_events"] = bench["roofline"]["avg_launch_ms"]
    out["algorithmic_bytes_per_launch"] = bench["roofline"]["algorithmic_bytes_per_launch"]
    out["device"] = bench.get("device")
    timed = bench["steps"] * bench["config"]["launches_per_step"]
elif steps:
    timed = steps * 2
stats = one("trace/*/*kernel_stats.csv")
if stats:
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
            out["all_dispatches"] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"]),
                                     "note": "rocprofv3 --stats over every dispatch, cold warm-up launches included"}
trace = one("trace/*/*kernel_trace.csv")
if trace:
    d = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(trace)) if r["Kernel_Name"].startswith("dr_")]
    d.sort()
    dur = [x[1] for x in d]
    if timed and len(dur) >= timed:
        dur = dur[-timed:]
    if dur:
        out["timed_dispatches"] = len(dur)
        out["timed_avg_ns"] = sum(dur) / len(dur)
        out["timed_min_ns"] = min(dur)
        out["timed_median_ns"] = statistics.median(dur)
        out["timed_max_ns"] = max(dur)
        out["avg_ns"] = out["timed_avg_ns"]
        if bench:
            out["timed_avg_over_hip_events"] = out["timed_avg_ns"] * 1e-6 / out["bench_avg_launch_ms_same_run_hip_events"]
for f in sorted(glob.glob(os.path.join(root, "*/*/*counter_collection.csv"))):
    acc = {}
    for r in csv.DictReader(open(f)):
        if not r["Kernel_Name"].startswith("dr_"):
            continue
        acc.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r.get("Start_Timestamp") and r.get("End_Timestamp"):
            # effective clock of THIS dispatch in THIS (profiled) pass: rocprofv3 reports the sum over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back)
            ns = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            if ns > 0:
                acc.setdefault("effective_clock_GHz_grbm_pass", []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"]) / 8.0 / ns))
                acc.setdefault("grbm_pass_launch_ns", []).append((int(r["Dispatch_Id"]), float(ns)))
        out.setdefault("vgpr", r["VGPR_Count"]); out.setdefault("lds", r["LDS_Block_Size"])
        out.setdefault("grid", r["Grid_Size"]); out.setdefault("wg", r["Workgroup_Size"])
    for k, v in acc.items():
        v.sort()
        vals = [x[1] for x in v]
        if timed and len(vals) >= timed:
            vals = vals[-timed:]
        out[k] = sum(vals) / len(vals)
failed = os.environ.get("FAILED_PASSES", "").split()
if failed:
    out["failed_passes"] = failed          # profile_bench.sh removed their directories: nothing of them (or of an older run) is summarised
if "FETCH_SIZE" in out and "WRITE_SIZE" in out and not ({"fetch", "write"} & set(failed)):
    out["fetch_bytes_corrected_x2"] = out["FETCH_SIZE"] * 1024 * 2
    out["write_bytes"] = out["WRITE_SIZE"] * 1024
    out["traffic_bytes_per_launch"] = out["fetch_bytes_corrected_x2"] + out["write_bytes"]
    if "algorithmic_bytes_per_launch" in out:
        out["traffic_over_algorithmic"] = out["traffic_bytes_per_launch"] / out["algorithmic_bytes_per_launch"]
if "GRBM_GUI_ACTIVE" in out and "effective_clock_GHz_grbm_pass" not in out and out.get("timed_avg_ns"):
    out["effective_clock_GHz_vs_trace_pass"] = out["GRBM_GUI_ACTIVE"] / 8.0 / out["timed_avg_ns"]     # cycles of one pass over the duration of another: +-3 %
if "TCC_HIT_sum" in out:
    out["l2_hit_rate"] = out["TCC_HIT_sum"] / (out["TCC_HIT_sum"] + out["TCC_MISS_sum"])
if "SQ_WAIT_ANY" in out and "SQ_WAVE_CYCLES" in out:
    out["wait_any_frac"] = out["SQ_WAIT_ANY"] / out["SQ_WAVE_CYCLES"]
if "SQ_WAVE_CYCLES" in out and out.get("GRBM_GUI_ACTIVE") and out.get("SQ_WAVES"):
    # SQ_WAVE_CYCLES counts resident waves in units of four cycles: the average number of waves a CU held, and the share of the launch
    # one wave lived (1/2: the grid ran as two rounds of workgroups)
    out["achieved_waves_per_cu"] = 4.0 * out["SQ_WAVE_CYCLES"] / (out["GRBM_GUI_ACTIVE"] / 8.0 * 256.0)
    out["wave_lifetime_frac"] = 4.0 * out["SQ_WAVE_CYCLES"] / out["SQ_WAVES"] / (out["GRBM_GUI_ACTIVE"] / 8.0)
if "SQ_LDS_BANK_CONFLICT" in out:
    out["lds_conflict_frac"] = out["SQ_LDS_BANK_CONFLICT"] / max(out["SQ_LDS_IDX_ACTIVE"], 1)
if "TCC_EA0_RDREQ_LEVEL_sum" in out and out.get("TCC_EA0_RDREQ_sum"):
    out["ea_read_latency_cycles"] = out["TCC_EA0_RDREQ_LEVEL_sum"] / out["TCC_EA0_RDREQ_sum"]
    out["ea_write_latency_cycles"] = out["TCC_EA0_WRREQ_LEVEL_sum"] / max(out["TCC_EA0_WRREQ_sum"], 1)
json.dump(out, open(os.path.join("gpurun_out", tag + "_counters.json"), "w"), indent=1)
keys = ("kernel", "timed_avg_ns", "timed_min_ns", "timed_median_ns", "bench_avg_launch_ms_same_run_hip_events", "timed_avg_over_hip_events", "traffic_over_algorithmic", "l2_hit_rate",
        "wait_any_frac", "lds_conflict_frac", "effective_clock_GHz_grbm_pass", "grbm_pass_launch_ns", "effective_clock_GHz_vs_trace_pass", "ea_read_latency_cycles", "ea_write_latency_cycles", "vgpr",
        "achieved_waves_per_cu", "wave_lifetime_frac")
print(json.dumps({k: out[k] for k in keys if k in out}))
