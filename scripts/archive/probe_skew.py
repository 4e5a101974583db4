#!/usr/bin/env python3
"""Launch time vs the distance between the input and the output array (scripts/probe_harness.py found 1.86 -> 1.60 ms for the fused
step-3 kernel when the output array starts 64 KiB off a 4 GiB multiple behind the input array).  One arena per workload, output at
input + array bytes (rounded up to 2 MiB) + delta; every bench headline / side kernel of the workload; 12 launches each.
Usage: probe_skew.py [workload ...]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import drstencil_amd as drs

K = 1024
DELTAS = [0, 2 * K, 4 * K, 8 * K, 12 * K, 16 * K, 32 * K, 64 * K, 68 * K, 96 * K, 128 * K, 192 * K, 256 * K, 320 * K, 512 * K, 576 * K, 1024 * K, 1088 * K, 2048 * K, 2112 * K, -64 * K, -192 * K]


def kernels_of(wl):
    out = [("headline", bench.TUNED[wl])]
    if wl in bench.STEP1:
        out.append(("step1", bench.STEP1[wl]))
    for i, o in enumerate(bench.FUSED3.get(wl, [])[:2]):
        out.append(("fusedN_%d" % i, o))
    return out


def main():
    wls = sys.argv[1:] or ["c4", "c3", "c2", "c5"]
    built = {}
    for wl in wls:
        w = bench.WORKLOADS[wl]
        built[wl] = [(n, drs.Kernel(list(o) + [w["stc"]])) for n, o in kernels_of(wl)]
    if os.environ.get("PROBE_BUILD_ONLY"):
        return
    import torch
    p = torch.cuda.get_device_properties(0)
    print("device uuid", getattr(p, "uuid", ""), flush=True)
    st = torch.cuda.current_stream().cuda_stream
    for wl in wls:
        ks = built[wl]
        i = ks[0][1].info
        pts = i["M"] * i["N"] * (i["L"] if i["ndim"] == 3 else 1)
        esz = 4 if i["dtype"] == "fp32" else 8
        nbytes = pts * esz
        span = (nbytes + (2 << 20) - 1) // (2 << 20) * (2 << 20)
        arena = torch.empty((2 * span + (8 << 20)) // 4, dtype=torch.float32, device="cuda")
        arena.uniform_()
        base = arena.data_ptr() + (4 << 20)
        print("== %s: %s, array %d MiB; columns: %s" % (wl, bench.WORKLOADS[wl]["name"], nbytes >> 20, ", ".join(n for n, _ in ks)), flush=True)
        for d in DELTAS:
            line = "   delta %8d B (%6.1f KiB):" % (d, d / 1024.0)
            for n, k in ks:
                nl, ms = k.run_timed(base, base + span + d, iterations=6 * 2 * k.info["step"], warmup=4, stream=st)
                line += "  %.4f" % (ms / nl)
            print(line, flush=True)
        del arena
        torch.cuda.empty_cache()


main()
