#!/usr/bin/env python3
"""Full-size parity margin of temporal-blocking kernels vs the fused gold kernel (GPU)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import drstencil_amd as drs
stc = os.path.join(ROOT, "benchmarks", "configs", "c4_3d7pt_star_1024.stc")
cfgs = {
    "t2": ["--3d", "--dtype", "fp32", "--step", "2", "--temporal", "1", "--prefetch", "--bx", "66", "--by", "15", "--block-merge-x", "4", "--block-merge-y", "2", "--sn", "32", "--xcd-remap", "0"],
    "t3": ["--3d", "--dtype", "fp32", "--step", "3", "--temporal", "1", "--prefetch", "--bx", "34", "--by", "15", "--block-merge-x", "4", "--block-merge-y", "3", "--sn", "48", "--xcd-remap", "0"],
}
kerns = {k: drs.Kernel(v + [stc]) for k, v in cfgs.items()}
import torch
for name, k in kerns.items():
    i = k.info
    shape = (i["L"], i["M"], i["N"])
    for seed, kind in ((1, "uniform[0,1)"), (2, "uniform[-1,1)")):
        g = torch.Generator(device="cuda").manual_seed(seed)
        A0 = torch.rand(shape, dtype=torch.float32, device="cuda", generator=g)
        if seed == 2:
            A0 = A0 * 2 - 1
        A = A0.clone(); B = torch.zeros_like(A); Ag = A0.clone(); Bg = torch.zeros_like(A)
        for iters in (4, 12):
            A.copy_(A0); B.zero_(); Ag.copy_(A0); Bg.zero_()
            n = k.run(A.data_ptr(), B.data_ptr(), iterations=iters)
            k.run(Ag.data_ptr(), Bg.data_ptr(), iterations=iters, gold=True)
            torch.cuda.synchronize()
            h = i["halo"]
            inner = tuple(slice(h, s - h) for s in shape)
            d = (A[inner] - Ag[inner]).abs()
            rel = (d / Ag[inner].abs().clamp_min(1e-30)).max().item()
            scale = Ag[inner].abs().max().item()
            print("%s %s iterations=%d launches=%d: max_rel %.3e  max_abs/scale %.3e" % (name, kind, iters, n, rel, d.max().item() / scale), flush=True)
        del A, B, Ag, Bg, A0
