"""Debugging aid for the round-2 fuzz mismatch: one launch of the failing kernel and of its gold kernel, cropped dump."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import drstencil_amd as drs
from mm_variants import variants
name, args = variants(ROOT)[0]
k = drs.Kernel(args)
import torch
info = k.info
shape = (info["L"], info["M"], info["N"])
g = torch.Generator(device="cuda").manual_seed(7)
A0 = torch.rand(shape, dtype=torch.float64, device="cuda", generator=g)
B, Bg = torch.zeros_like(A0), torch.zeros_like(A0)
k.launch(A0.data_ptr(), B.data_ptr()); k.launch_gold(A0.data_ptr(), Bg.data_ptr()); torch.cuda.synchronize()
d = (B != Bg).nonzero()
print("planes that differ after ONE launch:", sorted(set(d[:, 0].tolist())))
crop = (slice(0, 40), slice(0, 40), slice(0, 96))
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "r2z", "mm_dump.npz"), A0=A0[crop].cpu().numpy(), B=B[crop].cpu().numpy(), Bg=Bg[crop].cpu().numpy())
# run it a second time into a fresh buffer: deterministic?
B2 = torch.zeros_like(A0); k.launch(A0.data_ptr(), B2.data_ptr()); torch.cuda.synchronize()
print("second launch equals first:", bool(torch.equal(B, B2)))
# NaN-filled output: which points does the kernel never write?
B3 = torch.full_like(A0, float("nan")); k.launch(A0.data_ptr(), B3.data_ptr()); torch.cuda.synchronize()
nn = torch.isnan(B3).nonzero()
h = info["halo"]
inner = torch.isnan(B3[h:-h, h:-h, h:-h]).sum().item()
print("unwritten interior points:", inner)
