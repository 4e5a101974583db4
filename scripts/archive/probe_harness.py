#!/usr/bin/env python3
"""Why does explore.py time the fused step-3 kernel 15-20 % slower than bench.py does (and the step-2 headline 3-4 %)?  One process, one
device: the same three kernels timed (a) on torch.rand / zeros buffers as explore.py allocates them, (b) on buffers carved out of one
arena at controlled distances between the input and the output array, (c) in a sustained loop (clock settling).  Prints the device uuid."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import drstencil_amd as drs

STC = os.path.join(ROOT, "benchmarks", "configs", "c4_3d7pt_star_1024.stc")
K = {
    "s2_headline": "--3d --dtype fp32 --step 2 --prefetch --prefetch-depth 3 --bx 32 --by 16 --block-merge-x 4 --block-merge-y 2 --sn 32 --xcd-remap 2 --cc-opt -fno-slp-vectorize",
    "s3_64x8_sn64": "--3d --dtype fp32 --step 3 --prefetch --prefetch-depth 1 --block-merge-x 4 --order rows --pack 0 --cc-opt -fno-slp-vectorize --bx 64 --by 8 --block-merge-y 2 --sn 64 --xcd-remap 2",
    "s3_128x4_sn64": "--3d --dtype fp32 --step 3 --prefetch --prefetch-depth 1 --block-merge-x 4 --order rows --pack 0 --cc-opt -fno-slp-vectorize --bx 128 --by 4 --block-merge-y 2 --sn 64 --xcd-remap 2",
}


def main():
    kern = {n: drs.Kernel(a.split() + [STC]) for n, a in K.items()}
    if os.environ.get("PROBE_BUILD_ONLY"):
        return
    import torch
    p = torch.cuda.get_device_properties(0)
    print("device uuid", getattr(p, "uuid", ""), flush=True)
    n = 1024 ** 3
    st = torch.cuda.current_stream().cuda_stream

    def t(k, a, b, launches=8, warm=4):
        nl, ms = k.run_timed(a, b, iterations=(launches // 2) * 2 * k.info["step"], warmup=warm, stream=st)
        return ms / nl

    print("(a) torch.rand A / zeros B, separate allocations (explore.py)")
    A = torch.rand((1024, 1024, 1024), dtype=torch.float32, device="cuda")
    B = torch.zeros_like(A)
    print("    A %#x  B %#x  B - A = %d MiB + %d B" % (A.data_ptr(), B.data_ptr(), (B.data_ptr() - A.data_ptr()) >> 20, (B.data_ptr() - A.data_ptr()) & ((1 << 20) - 1)))
    for rnd in range(2):
        for name, k in kern.items():
            print("    %-14s %.4f ms" % (name, t(k, A.data_ptr(), B.data_ptr())), flush=True)
    del A, B
    torch.cuda.empty_cache()
    print("(b) one arena, output array at input + 4 GiB + delta")
    arena = torch.empty(2 * n + (64 << 20), dtype=torch.float32, device="cuda")
    arena.uniform_()
    base = arena.data_ptr()
    for delta in (0, 256, 1024, 4096, 65536, 1 << 20, (1 << 20) + 4096, 32 << 20, 128 << 20):
        a, b = base, base + 4 * n + delta
        line = "    delta %10d B:" % delta
        for name, k in kern.items():
            line += "  %s %.4f" % (name, t(k, a, b))
        print(line, flush=True)
    print("(c) sustained: 40 x 8 launches of s3_64x8_sn64, then of s2_headline (ms per launch, every 5th sample)")
    a, b = base, base + 4 * n
    for name in ("s3_64x8_sn64", "s2_headline"):
        v = [t(kern[name], a, b, warm=0) for _ in range(40)]
        print("    %-14s" % name, " ".join("%.3f" % x for x in v[::5]), flush=True)
    print("(d) bench.py's way: seeded planes")
    sys.path.insert(0, ROOT)
    import bench
    A = bench._seeded_planes(torch, 0, 1024, (1024, 1024), torch.float32, torch.device("cuda"))
    B = torch.zeros_like(A)
    print("    A %#x  B %#x" % (A.data_ptr(), B.data_ptr()))
    for name, k in kern.items():
        print("    %-14s %.4f ms" % (name, t(k, A.data_ptr(), B.data_ptr(), launches=20, warm=10)), flush=True)


main()
