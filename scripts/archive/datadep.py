#!/usr/bin/env python3
"""Does kernel time depend on the data (finite random vs overflowed inf/nan vs zeros)?  And how
repeatable is a configuration across fresh buffers?  (C4 kernels; builds before HIP init.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import drstencil_amd as drs
stc = os.path.join(ROOT, "benchmarks", "configs", "c4_3d7pt_star_1024.stc")
cfgs = {
    "temporal2_b66y15": ["--3d", "--dtype", "fp32", "--step", "2", "--temporal", "1", "--prefetch", "--bx", "66", "--by", "15", "--block-merge-x", "4", "--block-merge-y", "2", "--sn", "32", "--xcd-remap", "0"],
    "fused2_b32y16_x2": ["--3d", "--dtype", "fp32", "--step", "2", "--prefetch", "--bx", "32", "--by", "16", "--block-merge-x", "4", "--block-merge-y", "2", "--sn", "32", "--xcd-remap", "2"],
    "fused2_b64y8_sn24": ["--3d", "--dtype", "fp32", "--step", "2", "--prefetch", "--bx", "64", "--by", "8", "--block-merge-x", "4", "--block-merge-y", "2", "--sn", "24", "--xcd-remap", "2"],
    "step1_b64y4m4": ["--3d", "--dtype", "fp32", "--bx", "64", "--by", "4", "--block-merge-x", "4", "--block-merge-y", "4", "--sn", "8", "--xcd-remap", "2"],
}
kerns = {k: drs.Kernel(v + [stc]) for k, v in cfgs.items()}
import torch
shape = (1024, 1024, 1024)
A = torch.empty(shape, dtype=torch.float32, device="cuda"); B = torch.empty_like(A)
s = torch.cuda.current_stream().cuda_stream
def fill(kind):
    if kind == "random": A.uniform_(0, 1)
    elif kind == "zeros": A.zero_()
    elif kind == "inf": A.fill_(float("inf"))
    elif kind == "nan": A.fill_(float("nan"))
    elif kind == "const1": A.fill_(1.0)
    B.zero_()
for rnd in range(3):
    for kind in ("random", "zeros", "inf", "nan", "const1", "random"):
        for name, k in kerns.items():
            fill(kind)
            n, ms = k.run_timed(A.data_ptr(), B.data_ptr(), iterations=8 * k.info["step"], warmup=4, stream=s)
            print("round %d %-8s %-20s %.3f ms/launch %7.1f GStencil/s" % (rnd, kind, name, ms / n, k.updates_per_launch() * n / ms / 1e6), flush=True)
