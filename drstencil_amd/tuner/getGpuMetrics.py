#!/usr/bin/env python3
"""getGpuMetrics.py <config-name> -- scrape rocprofv3 output of one configuration into gpuMetrics.csv.

Counterpart of the reference's benchmarks/<stencil>/getGpuMetrics.py:4-38 (which pulls 58 Nsight Compute
metrics out of prof/<name>.csv).  Here the per-configuration metrics are the ones that matter for a
bandwidth-bound stencil on MI355X: kernel duration, HBM traffic from the FETCH_SIZE / WRITE_SIZE counters
(FETCH_SIZE doubled: gfx950 reports half of a wide coalesced stream), achieved GB/s against the 8 TB/s
roofline, launch geometry and register/LDS use.  `duration.log` gets the duration like the reference's.

Round 4: the "why" columns beside each winner (compile_run.sh's tcc / sq / sq2 / grbm passes), one counterpart per family of the reference's
row: L2 Hit Rate (its "L2 Hit Rate"), Effective Clock (its "SM Frequency": GRBM_GUI_ACTIVE / 8 XCDs / the dispatch's own duration),
Waves Waiting and Issue Stalled (its "No Eligible" / "Warp Cycles Per Issued Instruction"), Issue Busy ("Issue Slots Busy"), LDS Bank
Conflicts ("Mem Pipes Busy" has no closer twin), the instruction mix ("Executed Instructions", "Branch Instructions"), VALU Busy ("SM Busy"),
Waves and Occupancy ("Waves Per SM", "Theoretical Occupancy"), and the launch-limit family (launch_limits below).  A pass that did
not run leaves its cells empty."""
import csv
import glob
import os
import re
import sys

HEADER = ["Metric Name", "Duration", "Calls", "FETCH_SIZE", "WRITE_SIZE", "HBM Traffic", "Algorithmic Bytes", "Achieved Bandwidth",
          "Roofline Fraction", "GStencil/s", "Grid Size", "Block Size", "LDS Per Block", "VGPR", "SGPR", "Program Time", "RMS Error", "AGPR", "Scratch", "VGPR Spill",
          # round 4: why it is as fast as it is
          "Traffic / Algorithmic", "L2 Hit Rate", "Effective Clock", "Waves Waiting", "Issue Stalled", "Issue Busy", "LDS Bank Conflicts", "VALU Busy",
          "VALU Instructions", "VMEM Read Instructions", "VMEM Write Instructions", "LDS Instructions", "SALU Instructions", "Branch Instructions",
          "Waves", "Occupancy",
          # ... and figures derived from the above, named after the reference's columns where one exists
          "Memory Throughput", "Read Throughput", "Write Throughput", "Elapsed Cycles", "Executed Instructions", "Executed Ipc Elapsed", "Waves Per CU", "Threads",
          # ... the launch-limit family of the reference's row ("Block Limit Registers / Shared Mem / Warps", "Theoretical Active Warps per SM",
          # "Theoretical Occupancy", "Achieved Active Warps Per SM", "Achieved Occupancy"): what caps the resident workgroups of a CU
          # (512 VGPRs per SIMD lane in granules of 8, 160 KiB of LDS, 32 wave slots), and what the counters saw
          "Waves Per Workgroup", "Block Limit Registers", "Block Limit LDS", "Block Limit Waves", "Theoretical Active Waves Per CU", "Theoretical Occupancy",
          "Achieved Active Waves Per CU", "Achieved Occupancy", "Wave Lifetime"]
UNITS = ["", "nsecond", "", "KiB", "KiB", "byte", "byte", "GB/s", "of 8 TB/s", "", "", "", "byte", "", "", "ms", "", "", "byte/lane", "",
         "", "of L2 requests", "GHz", "of wave cycles", "of wave cycles", "of wave cycles", "of LDS cycles", "of wave cycles",
         "wave instructions", "wave instructions", "wave instructions", "wave instructions", "wave instructions", "wave instructions",
         "", "waves/SIMD",
         "GB/s (HBM, counters)", "GB/s", "GB/s", "cycle (busy, per XCD)", "wave instructions", "wave instructions / cycle / CU", "", "",
         "", "workgroups/CU", "workgroups/CU", "workgroups/CU", "", "of 32 wave slots", "", "of 32 wave slots", "of the launch"]


def _one(pattern):
    f = glob.glob(pattern, recursive=True)
    return f[0] if f else None


def _dr_rows(path):
    if not path:
        return []
    return [r for r in csv.DictReader(open(path)) if r.get("Kernel_Name", r.get("Name", "")).startswith("dr_")]


def launch_limits(wg, vgprs, agprs, lds, vals, cycles):
    """Resident workgroups per CU by resource, the occupancy they allow, and the occupancy the counters saw: SQ_WAVE_CYCLES counts
    resident waves in units of four cycles, so 4 x SQ_WAVE_CYCLES / (busy cycles x 256 CUs) is the average number of waves a CU held, and
    the same figure over SQ_WAVES and the cycles of the launch is how much of the launch one wave lived (1/2 = two rounds of workgroups)."""
    try:
        wpw = -(-int(wg) // 64)
    except ValueError:
        return [""] * 9
    regs = -(-(int(vgprs or 0) + int(agprs or 0)) // 8) * 8
    by_regs = (min(8, 512 // regs) * 4) // wpw if regs else ""
    by_lds = (160 * 1024) // int(lds) if str(lds).isdigit() and int(lds) > 0 else ""
    by_waves = 32 // wpw
    limit = min(x for x in (by_regs, by_lds, by_waves) if x != "")
    out = [wpw, by_regs, by_lds, by_waves, limit * wpw, limit * wpw / 32.0]
    if "SQ_WAVE_CYCLES" in vals and cycles:
        act = 4.0 * vals["SQ_WAVE_CYCLES"] / (cycles * 256.0)
        out += [act, act / 32.0, 4.0 * vals["SQ_WAVE_CYCLES"] / vals["SQ_WAVES"] / cycles if vals.get("SQ_WAVES") else ""]
    else:
        out += ["", "", ""]
    return out


def main(name=""):
    base = os.path.join("prof", name)
    stats = _dr_rows(_one(os.path.join(base, "trace", "**", "*_kernel_stats.csv")))
    dur = float(stats[0]["AverageNs"]) if stats else float("nan")
    calls = int(stats[0]["Calls"]) if stats else 0
    vals = {}
    meta = {}
    clock = float("nan")
    for key in ("fetch", "write", "tcc", "sq", "sq2", "grbm"):
        rows = _dr_rows(_one(os.path.join(base, key, "**", "*_counter_collection.csv")))
        if rows:
            meta = meta or rows[0]
            for cname in set(r["Counter_Name"] for r in rows):
                v = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == cname]
                vals[cname] = sum(v) / len(v)
            if key == "grbm":       # busy cycles over the SAME dispatch's duration (the counter sums the 8 XCDs)
                c = [float(r["Counter_Value"]) / 8.0 / max(1, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows
                     if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and "End_Timestamp" in r]
                clock = sum(c) / len(c) if c else clock
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
    def ratio(a, b):
        return vals[a] / vals[b] if a in vals and b in vals and vals[b] else ""
    occ = re.search(r"Occupancy \[waves/SIMD\]:\s*(\d+)", rep)
    why = [traffic / alg if alg and traffic == traffic else "",
           vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"]) if "TCC_HIT_sum" in vals and (vals["TCC_HIT_sum"] + vals.get("TCC_MISS_sum", 0)) else "",
           clock if clock == clock else "", ratio("SQ_WAIT_ANY", "SQ_WAVE_CYCLES"), ratio("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"), ratio("SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES"),
           ratio("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"), (vals["SQ_ACTIVE_INST_VALU"] / vals["SQ_WAVE_CYCLES"]) if "SQ_ACTIVE_INST_VALU" in vals and vals.get("SQ_WAVE_CYCLES") else "",
           vals.get("SQ_INSTS_VALU", ""), vals.get("SQ_INSTS_VMEM_RD", ""), vals.get("SQ_INSTS_VMEM_WR", ""), vals.get("SQ_INSTS_LDS", ""), vals.get("SQ_INSTS_SALU", ""),
           vals.get("SQ_INSTS_BRANCH", ""), vals.get("SQ_WAVES", ""), occ.group(1) if occ else ""]
    insts = [vals[k] for k in ("SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH") if k in vals]
    cycles = vals["GRBM_GUI_ACTIVE"] / 8.0 if "GRBM_GUI_ACTIVE" in vals else None
    try:
        threads = int(meta.get("Grid_Size", ""))
    except ValueError:
        threads = ""
    why += [traffic / dur if traffic == traffic and dur == dur and dur > 0 else "", fetch * 2048 / dur if fetch == fetch and dur == dur and dur > 0 else "",
            write * 1024 / dur if write == write and dur == dur and dur > 0 else "", cycles if cycles else "", sum(insts) if insts else "",
            sum(insts) / cycles / 256.0 if insts and cycles else "", vals["SQ_WAVES"] / 256.0 if "SQ_WAVES" in vals else "", threads]
    why += launch_limits(meta.get("Workgroup_Size", ""), _rep("VGPRs:"), _rep("AGPRs:"), meta.get("LDS_Block_Size", ""), vals, cycles)
    row = [name, dur, calls, fetch, write, traffic, alg, gbs, gbs / 8000.0, interior * mac.get("Step", 1) / dur if dur == dur and dur > 0 else float("nan"),
           meta.get("Grid_Size", ""), meta.get("Workgroup_Size", ""), meta.get("LDS_Block_Size", ""), meta.get("VGPR_Count", ""), meta.get("SGPR_Count", ""),
           t.group(1) if t else "", rms.group(1) if rms else "", _rep("AGPRs:"), _rep("ScratchSize [bytes/lane]:"), _rep("VGPRs Spill:")] + why
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
