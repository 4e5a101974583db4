#!/usr/bin/env python3
"""Follow-up of probe_cold.py: is the cold-array penalty of the fused step-3 kernel (first ~140 ms) a matter of TIME since the allocation or of
ACCESSES?  (1) fresh pair, 0.5 s of idle, then 12 x 4 launches; (2) fresh pair, no idle; (3) fresh input + warm output; (4) a fresh pair that
was first swept by 40 launches of the cheap step-1 kernel; (5) fresh pair filled by hipMemset-like zero_() only (no random fill)."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import drstencil_amd as drs


def main():
    w = bench.WORKLOADS["c4"]
    k = drs.Kernel(list(bench.FUSED3["c4"][0]) + [w["stc"]])
    k1 = drs.Kernel(list(bench.STEP1["c4"]) + [w["stc"]])
    if os.environ.get("PROBE_BUILD_ONLY"):
        return
    import torch
    print("device uuid", getattr(torch.cuda.get_device_properties(0), "uuid", ""), flush=True)
    st = torch.cuda.current_stream().cuda_stream

    def series(kk, a, b, reps, n=2):
        out = []
        for _ in range(reps):
            nl, ms = kk.run_timed(a, b, iterations=n * 2 * kk.info["step"], warmup=0, stream=st)
            out.append(ms / nl)
        return " ".join("%.3f" % x for x in out)

    def fresh():
        A = torch.rand((1024, 1024, 1024), dtype=torch.float32, device="cuda")
        B = torch.zeros_like(A)
        torch.cuda.synchronize()
        return A, B

    W, WB = fresh()
    print("(0) first pair of the process:            ", series(k, W.data_ptr(), WB.data_ptr(), 26), flush=True)
    A, B = fresh()
    time.sleep(0.5)
    print("(1) fresh pair after 0.5 s of idle:       ", series(k, A.data_ptr(), B.data_ptr(), 26), flush=True)
    del A, B; torch.cuda.empty_cache()
    A, B = fresh()
    print("(2) fresh pair, no idle:                  ", series(k, A.data_ptr(), B.data_ptr(), 26), flush=True)
    del A, B; torch.cuda.empty_cache()
    A, _ = fresh(); del _
    print("(3) fresh input, warm output:             ", series(k, A.data_ptr(), WB.data_ptr(), 26), flush=True)
    del A; torch.cuda.empty_cache()
    A, B = fresh()
    for _ in range(40):
        k1.launch(A.data_ptr(), B.data_ptr(), st); k1.launch(B.data_ptr(), A.data_ptr(), st)
    torch.cuda.synchronize()
    print("(4) fresh pair after 80 step-1 launches:  ", series(k, A.data_ptr(), B.data_ptr(), 12), flush=True)
    del A, B; torch.cuda.empty_cache()
    A = torch.empty((1024, 1024, 1024), dtype=torch.float32, device="cuda"); B = torch.empty_like(A)
    A.fill_(0.5); B.zero_(); torch.cuda.synchronize()
    print("(5) fresh pair, constant fill:            ", series(k, A.data_ptr(), B.data_ptr(), 26), flush=True)
    print("(6) the first pair again:                 ", series(k, W.data_ptr(), WB.data_ptr(), 6), flush=True)
    time.sleep(0.5)
    print("(7) constant-fill pair again (0.5 s later):", series(k, A.data_ptr(), B.data_ptr(), 6), flush=True)


main()
