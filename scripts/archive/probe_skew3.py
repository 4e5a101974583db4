#!/usr/bin/env python3
"""Dense sweep of the output array's position behind the input array inside ONE arena (C4: step-2 headline and fused step 3):
delta = k MiB + j * 64 KiB, k = 0..48, j in {0, 1, 3}; then the five best and five worst again (noise)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import drstencil_amd as drs

K, M, G = 1 << 10, 1 << 20, 1 << 30


def main():
    w = bench.WORKLOADS["c4"]
    ks = [("s2", drs.Kernel(list(bench.TUNED["c4"]) + [w["stc"]])), ("s3", drs.Kernel(list(bench.FUSED3["c4"][0]) + [w["stc"]]))]
    if os.environ.get("PROBE_BUILD_ONLY"):
        return
    import torch
    print("device uuid", getattr(torch.cuda.get_device_properties(0), "uuid", ""), flush=True)
    st = torch.cuda.current_stream().cuda_stream

    def t(k, a, b, n=4):
        nl, ms = k.run_timed(a, b, iterations=n * 2 * k.info["step"], warmup=2, stream=st)
        return ms / nl

    nb = 4 * G
    arena = torch.empty((2 * nb + 128 * M) // 4, dtype=torch.float32, device="cuda")
    arena.uniform_()
    base = (arena.data_ptr() + 8 * M) // (2 * M) * (2 * M)
    res = {}
    for k in range(0, 49):
        line = "k %2d MiB:" % k
        for j in (0, 1, 3):
            d = k * M + j * 64 * K
            a, b = t(ks[0][1], base, base + nb + d), t(ks[1][1], base, base + nb + d)
            res[d] = (a, b)
            line += "   +%3dK s2 %.3f s3 %.3f" % (j * 64, a, b)
        print(line, flush=True)
    order = sorted(res, key=lambda d: res[d][1])
    print("again (12 launches): five best / five worst for s3")
    for d in order[:5] + order[-5:]:
        print("   delta %9d B (%7.3f MiB): s2 %.4f s3 %.4f   (first pass %.3f %.3f)" % (d, d / M, t(ks[0][1], base, base + nb + d, 6), t(ks[1][1], base, base + nb + d, 6), res[d][0], res[d][1]), flush=True)
    order = sorted(res, key=lambda d: res[d][0])
    print("again: five best / five worst for s2")
    for d in order[:5] + order[-5:]:
        print("   delta %9d B (%7.3f MiB): s2 %.4f s3 %.4f   (first pass %.3f %.3f)" % (d, d / M, t(ks[0][1], base, base + nb + d, 6), t(ks[1][1], base, base + nb + d, 6), res[d][0], res[d][1]), flush=True)


main()
