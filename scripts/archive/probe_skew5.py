#!/usr/bin/env python3
"""One arena (in, out = in + 4 GiB + delta) against two separate allocations (out = its own allocation + delta): C4 step-2 headline,
step 1, fused step 3; warm buffers; forward and backward launches of the ping-pong timed separately; delta = 0 .. 56 MiB in 8 MiB steps."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import drstencil_amd as drs

M, G = 1 << 20, 1 << 30


def main():
    w = bench.WORKLOADS["c4"]
    ks = [("s2", drs.Kernel(list(bench.TUNED["c4"]) + [w["stc"]])), ("s1", drs.Kernel(list(bench.STEP1["c4"]) + [w["stc"]])), ("s3", drs.Kernel(list(bench.FUSED3["c4"][0]) + [w["stc"]]))]
    if os.environ.get("PROBE_BUILD_ONLY"):
        return
    import torch
    print("device uuid", getattr(torch.cuda.get_device_properties(0), "uuid", ""), flush=True)
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def one_way(k, a, b, n=6):
        k.launch(a, b, stream=st.cuda_stream)
        e0.record(st)
        for _ in range(n):
            k.launch(a, b, stream=st.cuda_stream)
        e1.record(st)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    def warm(a, b):
        tw = time.perf_counter()
        while time.perf_counter() - tw < 0.5:
            for _, k in ks:
                k.launch(a, b, stream=st.cuda_stream); k.launch(b, a, stream=st.cuda_stream)
            torch.cuda.synchronize()

    nb = 4 * G
    for mode in ("arena", "separate", "separate-again"):
        if mode == "arena":
            arena = torch.empty((2 * nb + 80 * M) // 4, dtype=torch.float32, device="cuda"); arena.uniform_()
            a, b0 = arena.data_ptr(), arena.data_ptr() + nb
            keep = (arena,)
        else:
            A = torch.rand(nb // 4, dtype=torch.float32, device="cuda")
            Bx = torch.zeros((nb + 80 * M) // 4, dtype=torch.float32, device="cuda")
            a, b0 = A.data_ptr(), Bx.data_ptr()
            keep = (A, Bx)
        warm(a, b0 + 32 * M)
        print("== %s: in %#x, out base %#x" % (mode, a, b0), flush=True)
        for d in range(0, 64, 8):
            line = "   out base + %2d MiB:" % d
            tot = 0.0
            for n, k in ks:
                f, r = one_way(k, a, b0 + d * M), one_way(k, b0 + d * M, a)
                line += "   %s fwd %.4f bwd %.4f" % (n, f, r)
            print(line, flush=True)
        del keep
        if mode == "arena":
            del arena
        else:
            del A, Bx
        torch.cuda.empty_cache()


main()
