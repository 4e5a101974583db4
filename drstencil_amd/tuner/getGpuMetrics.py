#!/usr/bin/env python3
"""getGpuMetrics.py <config-name> -- scrape rocprofv3 output of one configuration into gpuMetrics.csv.

Counterpart of the reference's benchmarks/<stencil>/getGpuMetrics.py:4-38 (which pulls 58 Nsight Compute
metrics out of prof/<name>.csv).  Here the per-configuration metrics are the ones that matter for a
bandwidth-bound stencil on MI355X: kernel duration, HBM traffic from the FETCH_SIZE / WRITE_SIZE counters
(FETCH_SIZE doubled: gfx950 reports half of a wide coalesced stream), achieved GB/s against the 8 TB/s
roofline, launch geometry and register/LDS use.  `duration.log` gets the duration like the reference's."""
import csv
import glob
import os
import re
import sys

HEADER = ["Metric Name", "Duration", "Calls", "FETCH_SIZE", "WRITE_SIZE", "HBM Traffic", "Algorithmic Bytes", "Achieved Bandwidth",
          "Roofline Fraction", "GStencil/s", "Grid Size", "Block Size", "LDS Per Block", "VGPR", "SGPR", "Program Time", "RMS Error", "AGPR", "Scratch", "VGPR Spill"]
UNITS = ["", "nsecond", "", "KiB", "KiB", "byte", "byte", "GB/s", "of 8 TB/s", "", "", "", "byte", "", "", "ms", "", "", "byte/lane", ""]


def _one(pattern):
    f = glob.glob(pattern, recursive=True)
    return f[0] if f else None


def _dr_rows(path):
    if not path:
        return []
    return [r for r in csv.DictReader(open(path)) if r.get("Kernel_Name", r.get("Name", "")).startswith("dr_")]


def main(name=""):
    base = os.path.join("prof", name)
    stats = _dr_rows(_one(os.path.join(base, "trace", "**", "*_kernel_stats.csv")))
    dur = float(stats[0]["AverageNs"]) if stats else float("nan")
    calls = int(stats[0]["Calls"]) if stats else 0
    vals = {}
    meta = {}
    for key in ("fetch", "write"):
        rows = _dr_rows(_one(os.path.join(base, key, "**", "*_counter_collection.csv")))
        if rows:
            meta = rows[0]
            for cname in set(r["Counter_Name"] for r in rows):
                v = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == cname]
                vals[cname] = sum(v) / len(v)
    src = open(os.path.join("cu", name + ".hip")).read() if os.path.exists(os.path.join("cu", name + ".hip")) else ""
    mac = {m.group(1): int(m.group(2)) for m in re.finditer(r"^#define (L|M|N|Halo|Step) (-?\d+)", src, re.M)}
    esz = 4 if "typedef float real_t" in src else 8
    npts = mac.get("M", 0) * mac.get("N", 0) * (mac.get("L", 1) if "#define L " in src else 1)
    alg = 2 * esz * npts
    fetch, write = vals.get("FETCH_SIZE", float("nan")), vals.get("WRITE_SIZE", float("nan"))
    traffic = fetch * 1024 * 2 + write * 1024
    gbs = alg / dur if dur == dur and dur > 0 else float("nan")
    h = mac.get("Halo", 0)
    interior = 1
    for d in ([mac.get("L", 1)] if "#define L " in src else []) + [mac.get("M", 0), mac.get("N", 0)]:
        interior *= max(d - 2 * h, 0)
    log = open(base + ".log").read() if os.path.exists(base + ".log") else ""
    t = re.search(r"GPU computation time: ([0-9.]+) ms", log)
    rms = re.search(r"\[Test\] RMS Error: (\S+)", log)
    # the compiler's resource report of the dr_ kernel (compile_run.sh keeps it)
    rep = open(base + ".resources.txt").read() if os.path.exists(base + ".resources.txt") else ""
    at = rep.find("Function Name: dr_")
    rep = rep[at:rep.find("Function Name:", at + 14) if at >= 0 and rep.find("Function Name:", at + 14) > 0 else None] if at >= 0 else ""
    def _rep(label):
        m = re.search(re.escape(label) + r"\s*(\d+)", rep)
        return m.group(1) if m else ""
    row = [name, dur, calls, fetch, write, traffic, alg, gbs, gbs / 8000.0, interior * mac.get("Step", 1) / dur if dur == dur and dur > 0 else float("nan"),
           meta.get("Grid_Size", ""), meta.get("Workgroup_Size", ""), meta.get("LDS_Block_Size", ""), meta.get("VGPR_Count", ""), meta.get("SGPR_Count", ""),
           t.group(1) if t else "", rms.group(1) if rms else "", _rep("AGPRs:"), _rep("ScratchSize [bytes/lane]:"), _rep("VGPRs Spill:")]
    new = not os.path.exists("gpuMetrics.csv")
    with open("gpuMetrics.csv", "a", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_ALL)
        if new:
            w.writerow(HEADER)
            w.writerow(UNITS)
        w.writerow(row)
    with open("duration.log", "a") as f:
        f.write(str(dur) + "\n")


if __name__ == "__main__":
    main(sys.argv[1])
