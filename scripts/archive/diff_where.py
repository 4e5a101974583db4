#!/usr/bin/env python3
"""Debugging aid: run one configuration's dr_ kernel and its gold kernel on the same seeded input on the GPU and say WHERE
they differ (index ranges per dimension, the first few points, tile coordinates).
usage: diff_where.py <drstencil arguments ... file.stc>   (build the kernel before; DRS_KCACHE picks the cache)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import drstencil_amd as drs


def main():
    k = drs.Kernel(sys.argv[1:])
    import torch
    info = k.info
    shape = (info["L"], info["M"], info["N"]) if info["ndim"] == 3 else (info["M"], info["N"])
    dt = torch.float32 if info["dtype"] == "fp32" else torch.float64
    g = torch.Generator(device="cuda").manual_seed(7)
    A0 = torch.rand(shape, dtype=dt, device="cuda", generator=g)
    A, B, Ag, Bg = A0.clone(), torch.zeros_like(A0), A0.clone(), torch.zeros_like(A0)
    k.run(A.data_ptr(), B.data_ptr())
    k.run(Ag.data_ptr(), Bg.data_ptr(), gold=True)
    torch.cuda.synchronize()
    print(json.dumps({x: info.get(x) for x in ("name", "L", "M", "N", "halo", "step", "taps", "threads", "grid", "lds_bytes", "schedule", "stage", "resident_planes", "carry")}))
    for name, X, G in (("A", A, Ag), ("B", B, Bg)):
        d = (X != G).nonzero()
        if d.numel() == 0:
            print(name, "identical"); continue
        print(name, "%d points differ" % d.shape[0])
        for ax in range(d.shape[1]):
            v = d[:, ax]
            print("  axis %d: min %d max %d, distinct %s" % (ax, int(v.min()), int(v.max()), sorted(set(v.tolist()))[:40]))
        for p in d[:8].tolist():
            print("  at", p, "dr", float(X[tuple(p)]), "gold", float(G[tuple(p)]))


if __name__ == "__main__":
    main()
