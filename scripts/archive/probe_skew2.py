#!/usr/bin/env python3
"""Follow-up of probe_skew.py: which relation between the input and the output array is fast?  (a) one arena, output at input + 4 GiB
+ delta for coarse deltas up to 3 GiB; (b) separate allocations with fillers of varying size in between (addresses printed).
C4 kernels: step-2 headline, fused step 3 (64x8 lanes)."""
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

    def t(k, a, b):
        nl, ms = k.run_timed(a, b, iterations=6 * 2 * k.info["step"], warmup=4, stream=st)
        return ms / nl

    nb = 4 * G
    arena = torch.empty((2 * nb + 3 * G + 64 * M) // 4, dtype=torch.float32, device="cuda")
    arena.uniform_()
    base = (arena.data_ptr() + 32 * M) // (2 * M) * (2 * M)
    print("(a) arena %#x, input at %#x, output at input + 4 GiB + delta" % (arena.data_ptr(), base))
    for d in [0, 64 * K, 4 * M, 4 * M + 64 * K, 8 * M, 16 * M, 16 * M + 64 * K, 32 * M, 64 * M, 128 * M, 256 * M, 256 * M + 64 * K, 512 * M, 1 * G, 1 * G + 64 * K, 2 * G, 2 * G + 64 * K, 3 * G,
              3 * G + 192 * K]:
        print("   delta %11d B (%8.2f MiB):  s2 %.4f  s3 %.4f" % (d, d / M, t(ks[0][1], base, base + nb + d), t(ks[1][1], base, base + nb + d)), flush=True)
    print("(a2) input moved inside the arena (output at input + 4 GiB + 64 KiB)")
    for off in [0, 4 * K, 64 * K, 1 * M, 2 * M + 4 * K, 5 * M]:
        print("   input at base + %8d B:  s2 %.4f  s3 %.4f" % (off, t(ks[0][1], base + off, base + off + nb + 64 * K), t(ks[1][1], base + off, base + off + nb + 64 * K)), flush=True)
    del arena
    torch.cuda.empty_cache()
    print("(b) separate allocations, a filler of f MiB allocated between them")
    for f in [0, 2, 6, 34, 130, 514, 1026, 2050, 3, 70, 1000]:
        A = torch.rand((1024, 1024, 1024), dtype=torch.float32, device="cuda")
        F = torch.empty(max(1, f) * M, dtype=torch.uint8, device="cuda") if f else None
        B = torch.zeros_like(A)
        print("   filler %5d MiB: A %#x B %#x (B - A = %+.1f MiB):  s2 %.4f  s3 %.4f" % (f, A.data_ptr(), B.data_ptr(), (B.data_ptr() - A.data_ptr()) / M,
                                                                                      t(ks[0][1], A.data_ptr(), B.data_ptr()), t(ks[1][1], A.data_ptr(), B.data_ptr())), flush=True)
        del A, B, F
        torch.cuda.empty_cache()


main()
