import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from helpers import write_stc
import drstencil_amd as drs
STAR3 = [(0, 0, 0, 0.3), (1, 0, 0, 0.2), (-1, 0, 0, 0.1), (0, 1, 0, 0.15), (0, -1, 0, 0.05), (0, 0, 1, 0.12), (0, 0, -1, 0.08)]
BASE = "--3d --dtype fp64 --bx 16 --by 8 --sn 16 --stream-unroll 4 --step 3 --dist 3 --block-merge-x 2 --block-merge-y 2 --merge-forward 5 --xrim dpp --xcd-remap 2 --schedule scatter --uniform-loads 1 --store-mask buffer --stage dma".split()
def variants(root="/root/repo"):
    out = []
    d = os.path.join(root, "gpurun_out", "fuzz_gold"); os.makedirs(d, exist_ok=True)
    big = os.path.join(d, "g3_star.stc"); small = os.path.join(d, "g3_small.stc")
    write_stc(big, 3, (90, 150, 2100), 4, STAR3); write_stc(small, 3, (30, 50, 300), 4, STAR3)
    for line in open(os.path.join(root, "scripts", "exp_mm_variants.txt")):
        o = list(BASE)
        for e in line.split():
            if e.startswith("-rm:"):
                i = o.index(e[4:]); del o[i:i + 2]
            elif e.startswith("+add:"):
                o.append(e[5:])
            elif e.startswith("-set:"):
                k, v = e[5:].split("="); o[o.index(k) + 1] = v
        for stc in ((big, small) if os.environ.get("MM_SMALL") else (big,)):
            out.append((line.strip() + (" [small]" if stc == small else ""), o + [stc]))
    return out
if __name__ == "__main__":
    if sys.argv[1] == "build":
        from concurrent.futures import ProcessPoolExecutor
        def b(v):
            try: drs.Kernel(v[1]); return v[0] + ": built"
            except Exception as e: return v[0] + ": " + " ".join(str(e).split())[:200]
        with ProcessPoolExecutor(max_workers=6) as ex:
            for r in ex.map(b, variants()): print(r, flush=True)
    else:
        import torch
        for name, args in variants():
            try: k = drs.Kernel(args)
            except Exception as e: print(name, "-> not available:", " ".join(str(e).split())[:120]); continue
            info = k.info
            shape = (info["L"], info["M"], info["N"])
            dt = torch.float32 if info["dtype"] == "fp32" else torch.float64
            g = torch.Generator(device="cuda").manual_seed(7)
            A0 = torch.rand(shape, dtype=dt, device="cuda", generator=g)
            A, B, Ag, Bg = A0.clone(), torch.zeros_like(A0), A0.clone(), torch.zeros_like(A0)
            k.run(A.data_ptr(), B.data_ptr()); k.run(Ag.data_ptr(), Bg.data_ptr(), gold=True); torch.cuda.synchronize()
            d = (B != Bg).nonzero()
            print(name, "->", "identical" if d.numel() == 0 and torch.equal(A, Ag) else "%d points of B differ, planes %s" % (d.shape[0], sorted(set(d[:, 0].tolist()))[:12]), flush=True)
