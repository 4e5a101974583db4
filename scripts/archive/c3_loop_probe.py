#!/usr/bin/env python3
"""c3_loop_probe.py -- why bench.py's per-step loop reports a longer average launch on C3 (0.2 ms launches) than one C loop."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, drstencil_amd as drs
w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
k = drs.Kernel(bench.TUNED[sys.argv[1] if len(sys.argv) > 1 else "c3"] + [w["stc"]])
import torch
spec = drs.Spec(w["stc"], w["ndim"], 2)
L, M, N = spec.dims
A = torch.rand((L, M, N), device="cuda"); B = torch.zeros_like(A)
s = torch.cuda.current_stream()
for _ in range(5): k.run(A.data_ptr(), B.data_ptr(), stream=s.cuda_stream)
torch.cuda.synchronize()
def ev(fn, nl):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record(s); fn(); th = time.perf_counter() - t0; e1.record(s); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / nl * 1e3, th / nl * 1e6
for rep in range(3):
    a = ev(lambda: [k.run(A.data_ptr(), B.data_ptr(), stream=s.cuda_stream) for _ in range(40)], 80)
    b = ev(lambda: k.run(A.data_ptr(), B.data_ptr(), iterations=160, stream=s.cuda_stream), 80)
    n, ms = k.run_timed(A.data_ptr(), B.data_ptr(), iterations=160, warmup=0, stream=s.cuda_stream)
    print("40 python calls x 2 launches: %.1f us/launch (host %.1f us/launch) | one C loop of 80: %.1f us/launch (host %.1f) | run_timed: %.1f us/launch" % (a[0], a[1], b[0], b[1], ms / n * 1e3), flush=True)
