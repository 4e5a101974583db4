#!/usr/bin/env python3
"""The first ~0.2 s of launches after a buffer is allocated run 15-20 % slower for the fused step-3 kernel (scripts/probe_skew3.py: 1.92 -> 1.62 ms
with nothing else changed).  Is it the GPU (clocks after an idle period) or the memory (fresh pages)?  C4, fused step 3 + step-2 headline:
 (1) fresh buffers, 30 x 4 launches back to back;  (2) the same warm buffers after 2 s of idle;  (3) fresh buffers allocated while the GPU is kept
 busy on the old ones, first launches timed immediately;  (4) warm buffers, after torch.cuda.empty_cache() + a 4 GiB allocation and free."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import drstencil_amd as drs


def main():
    w = bench.WORKLOADS["c4"]
    ks = {"s2": drs.Kernel(list(bench.TUNED["c4"]) + [w["stc"]]), "s3": drs.Kernel(list(bench.FUSED3["c4"][0]) + [w["stc"]])}
    if os.environ.get("PROBE_BUILD_ONLY"):
        return
    import torch
    print("device uuid", getattr(torch.cuda.get_device_properties(0), "uuid", ""), flush=True)
    st = torch.cuda.current_stream().cuda_stream

    def series(k, a, b, reps, n=2):
        out = []
        for _ in range(reps):
            nl, ms = k.run_timed(a, b, iterations=n * 2 * k.info["step"], warmup=0, stream=st)
            out.append(ms / nl)
        return " ".join("%.3f" % x for x in out)

    for name in ("s3", "s2"):
        k = ks[name]
        print("== %s" % name)
        A = torch.rand((1024, 1024, 1024), dtype=torch.float32, device="cuda")
        B = torch.zeros_like(A)
        torch.cuda.synchronize()
        print("(1) fresh buffers, 30 x 4 launches:      ", series(k, A.data_ptr(), B.data_ptr(), 30), flush=True)
        time.sleep(2.0)
        print("(2) same buffers after 2 s idle:          ", series(k, A.data_ptr(), B.data_ptr(), 12), flush=True)
        # keep the GPU busy on the warm pair while a fresh pair is allocated and filled
        for _ in range(40):
            k.launch(A.data_ptr(), B.data_ptr())
        A2 = torch.rand((1024, 1024, 1024), dtype=torch.float32, device="cuda")
        B2 = torch.zeros_like(A2)
        for _ in range(40):
            k.launch(A.data_ptr(), B.data_ptr())
        print("(3) fresh pair right behind 80 warm launches:", series(k, A2.data_ptr(), B2.data_ptr(), 20), flush=True)
        print("    ... and the old pair again:             ", series(k, A.data_ptr(), B.data_ptr(), 6), flush=True)
        del A2, B2
        torch.cuda.empty_cache()
        X = torch.empty(1 << 30, dtype=torch.float32, device="cuda"); X.zero_(); del X
        torch.cuda.empty_cache()
        print("(4) old pair after an unrelated alloc/free: ", series(k, A.data_ptr(), B.data_ptr(), 8), flush=True)
        # (5) only the output is fresh / only the input is fresh
        B3 = torch.zeros_like(A)
        print("(5) warm input, fresh output:              ", series(k, A.data_ptr(), B3.data_ptr(), 16), flush=True)
        del A, B, B3
        torch.cuda.empty_cache()


main()
