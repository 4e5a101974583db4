#!/usr/bin/env python3
"""repro_gpu.py -- the two kernels the round-1 parity fuzz found miscompiled (both spill to scratch; prefetch depth 2 was the
automatic choice then), with option and -- via DRS_HIPCC=scripts/hipcc_wrappers/<x>.sh -- compiler-flag variants, against the oracle.
Needs DRS_ALLOW_SCRATCH=1 (the runtime refuses such kernels now)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import drstencil_amd as drs

BASE = {
    "c": ("t3_cross", "--3d --dtype fp64 --bx 16 --by 8 --sn 4 --stream-unroll 4 --step 3 --dist 6 --block-merge-x 2 --block-merge-y 2 --merge-forward 5 --prefetch --prefetch-depth 2 --xrim dpp --xcd-remap 2"),
    "o": ("t3_odd", "--3d --dtype fp64 --bx 32 --by 8 --sn 32 --stream-unroll 4 --step 3 --dist 3 --block-merge-x 2 --block-merge-y 2 --merge-forward 5 --prefetch --prefetch-depth 2 --xrim dpp --xcd-remap 2 --temporal 1"),
}
VARIANTS = [""] if os.environ.get("ONLY_BASE") else ["", "--prefetch-depth 1", "--prefetch-depth 3", "--prefetch-auto 0", "--xrim lds", "--nt-store 0", "--clamp-loads 0", "--xcd-remap 0", "--schedule window"]


def main():
    jobs = []
    for key, (stc, opts) in BASE.items():
        for v in VARIANTS:
            o = opts.split()
            for tok in v.split()[::2]:
                if tok in o:
                    i = o.index(tok); del o[i:i + 2]
            jobs.append((key, v, stc, o + v.split()))
    kerns = []
    for key, v, stc, o in jobs:
        try:
            kerns.append((key, v, stc, o, drs.Kernel(o + [os.path.join(ROOT, "tests", "stc", stc + ".stc")])))
        except Exception as e:
            print(key, v, "BUILD FAILED", str(e)[-100:])
    import torch
    import oracle
    for key, v, stc, o, k in kerns:
        path = os.path.join(ROOT, "tests", "stc", stc + ".stc")
        spec = oracle.Spec(path, 3, 3)
        A0 = oracle.fill_random(spec.shape, np.float64)
        Ar, Br = A0.copy(), np.zeros_like(A0)
        oracle.run(spec, Ar, Br, contract=1)
        res = []
        for rep in range(3):
            dA = torch.from_numpy(A0).cuda(); dB = torch.zeros_like(dA)
            k.run(dA.data_ptr(), dB.data_ptr()); torch.cuda.synchronize()
            A, B = dA.cpu().numpy(), dB.cpu().numpy()
            bad = np.argwhere((A != Ar) | (B != Br))
            res.append(len(bad))
        desc = ""
        if res[-1]:
            desc = " first %s last %s maxabs %.3g" % (bad.min(0), bad.max(0), max(np.abs(A - Ar).max(), np.abs(B - Br).max()))
        if res[-1] and os.environ.get("DETAIL"):
            bz, by_, bx = bad[:, 0], bad[:, 1], bad[:, 2]
            print("   z hist", dict(zip(*np.unique(bz, return_counts=True))))
            print("   y hist", dict(zip(*np.unique(by_, return_counts=True))))
            ux, cx = np.unique(bx, return_counts=True)
            print("   x values (first 40)", list(zip(ux[:40], cx[:40])), "n distinct x", len(ux))
            wrongA = (A != Ar); print("   A wrong", wrongA.sum(), "B wrong", (B != Br).sum())
        print("%s %-22s stages %d un %s lds %6d : mismatching points per run %s%s" % (key, v or "(as found)", k.info.get("stages", 1), k.info.get("unroll"), k.info["lds_bytes"], res, desc), flush=True)


if __name__ == "__main__":
    main()
