#!/usr/bin/env python3
"""Randomised GPU sweep on LARGER ragged grids (several tiles in every dimension, x wider than any workgroup tile):
random configurations from the tuner's space, the optimised kernel against the gold kernel of the same emitted source
(the gold kernel's arithmetic is pinned to the oracle by tests/test_gpu_parity.py) -- bit for bit for single-pass
kernels, within the dtype's bar for temporal pipelines.  usage: fuzz_gold.py <n> <seed>.  Builds before HIP is up."""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import drstencil_amd as drs
from drstencil_amd.tuner import tuning as t
from helpers import write_stc
from concurrent.futures import ProcessPoolExecutor

STAR3 = [(0, 0, 0, 0.3), (1, 0, 0, 0.2), (-1, 0, 0, 0.1), (0, 1, 0, 0.15), (0, -1, 0, 0.05), (0, 0, 1, 0.12), (0, 0, -1, 0.08)]
CROSS3 = [(0, 0, 0, 0.3)] + [(a, b, c, 0.0875) for a in (-1, 1) for b in (-1, 1) for c in (-1, 1)]
STAR2 = [(0, 0, 0.3), (0, 1, 0.2), (1, 0, 0.15), (0, -1, 0.25), (-1, 0, 0.1)]
BOX25 = [(j, i, 0.04 + 0.001 * (5 * j + i)) for j in range(-2, 3) for i in range(-2, 3)]
SPECS = [(3, "g3_star", (90, 150, 2100), STAR3, 1, False), (3, "g3_cross", (70, 130, 1540), CROSS3, 1, True),
         (2, "g2_star", (1, 3000, 5000), STAR2, 1, False), (2, "g2_box25", (1, 2100, 4100), BOX25, 2, False)]
if os.environ.get("FUZZ_GOLD_BIG"):   # BASELINE sizes: offsets beyond 2^31 bytes, thousands of tiles (fp32 3D / fp64 2D only, to bound the run)
    SPECS = [(3, "b3_star_1024", (1024, 1024, 1024), STAR3, 1, False), (2, "b2_box25_16384", (1, 16384, 16384), BOX25, 2, False),
             (2, "b2_star_8192", (1, 8192, 8192), STAR2, 1, False)]
    DTYPES = {"b3_star_1024": ("fp32",), "b2_box25_16384": ("fp64",), "b2_star_8192": ("fp32",)}
else:
    DTYPES = {}


def build(job):
    try:
        drs.Kernel(job[2]); return None
    except Exception as e:
        return "%s: %s" % (" ".join(job[2][:-1]), " ".join(str(e).split())[:200])


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    random.seed(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    out = os.path.join(ROOT, "gpurun_out", "fuzz_gold"); os.makedirs(out, exist_ok=True)
    jobs = []
    for ndim, name, dims, pts, order, cross in SPECS:
        stc = os.path.join(out, name + ".stc")
        write_stc(stc, ndim, dims, 4, pts)
        for dtype in DTYPES.get(name, ("fp32", "fp64")):
            t.order, t.ndim, t.elem_bytes = order, ndim, 4 if dtype == "fp32" else 8
            space = t.enumerate_space((1, 2, 3) if order == 1 and not cross else (1, 2))
            for v in random.sample(space, min(len(space), max(1, n // (len(DTYPES.get(name, (1, 2))) * len(SPECS))))):
                cl = t.cfgToCommandLine(v).split()
                if cross:
                    i = cl.index("--dist"); cl[i + 1] = str(2 * v[0])
                if "--prefetch-depth" in cl:
                    cl[cl.index("--prefetch-depth") + 1] = str(random.choice([1, 2, 3]))
                # round-2 knobs: --merge-forward of the reuse schedule, the memory path, LDS-DMA staging
                if "--schedule" not in cl and random.random() < 0.6:
                    cl[cl.index("--merge-forward") + 1] = str(random.choice([0, 2, 3, 100]))
                if random.random() < 0.25:
                    cl += ["--uniform-loads", str(random.choice([1, 2]))]
                if random.random() < 0.25:
                    cl += ["--store-mask", "buffer"]
                if random.random() < 0.15:
                    cl += ["--drain", str(random.choice([1, 2]))]
                if random.random() < 0.3 and "--temporal" not in cl and "--cyclic-merge-y" not in cl and (ndim == 3 or "--streaming" in cl):
                    cl += ["--stage", "dma"]
                if random.random() < 0.3:
                    cl += ["--defer-stores", "1"]
                if os.environ.get("FUZZ_ROUND4", "1") != "0":     # rounds 3 and 4's knobs on top (drawn last): rows order / pinned sums / ...; skewed pipelines, XCD maps, strided x merge
                    import fuzz_parity as fp
                    fp.round3_knobs(random, cl)
                    fp.round4_knobs(random, cl)
                    if "--skew" in cl and ndim == 2 and "--streaming" not in cl:
                        del cl[cl.index("--skew"):cl.index("--skew") + 2]
                jobs.append((ndim, dtype, (["--3d"] if ndim == 3 else []) + ["--dtype", dtype] + cl + [stc], dims))
    t0 = time.time()
    with ProcessPoolExecutor(max_workers=int(os.environ.get("FUZZ_JOBS", "16"))) as ex:
        errs = list(ex.map(build, jobs))
    ok = [j for j, e in zip(jobs, errs) if e is None]
    rej = [e for e in errs if e is not None]
    print("built %d kernels in %.0f s; %d rejected or refused (%d for scratch spills)" % (len(ok), time.time() - t0, len(rej), sum("scratch" in e for e in rej)), flush=True)
    if os.environ.get("FUZZ_BUILD_ONLY"):   # fill the kernel cache on a box without a GPU; the GPU run then finds every kernel built
        return
    kerns = [(j, drs.Kernel(j[2])) for j in ok]
    import torch
    bad = exact = 0
    worst = {"fp32": 0.0, "fp64": 0.0}
    for cnt, ((ndim, dtype, args, dims), k) in enumerate(kerns, 1):
        if cnt % 50 == 0:
            print("... %d / %d checked, %d mismatches" % (cnt, len(kerns), bad), flush=True)      # a long silent run looks hung to the GPU pool
        tdt = torch.float32 if dtype == "fp32" else torch.float64
        shape = dims if ndim == 3 else dims[1:]
        g = torch.Generator(device="cuda").manual_seed(7)
        A0 = torch.rand(shape, dtype=tdt, device="cuda", generator=g)
        A, B = A0.clone(), torch.zeros_like(A0)
        Ag, Bg = A0.clone(), torch.zeros_like(A0)
        k.run(A.data_ptr(), B.data_ptr())
        k.run(Ag.data_ptr(), Bg.data_ptr(), gold=True)
        torch.cuda.synchronize()
        if k.info.get("stages", 1) > 1:
            h = k.info["halo"]
            inner = tuple(slice(h, s - h) for s in shape)
            rel = max(float(((A - Ag)[inner].abs() / Ag[inner].abs().clamp_min(1e-30)).max()), float(((B - Bg)[inner].abs() / Bg[inner].abs().clamp_min(1e-30)).max()))
            worst[dtype] = max(worst[dtype], rel)
            good = rel <= (1e-6 if dtype == "fp32" else 1e-12)
        else:
            good = bool(torch.equal(A, Ag) and torch.equal(B, Bg))
            exact += good
        if not good:
            bad += 1
            print("MISMATCH", " ".join(args[:-1]), os.path.basename(args[-1]), flush=True)
        del A, B, Ag, Bg, A0
    print("%d configurations checked against their gold kernel: %d single-pass bit-exact, %d temporal within tolerance (worst fp32 %.3g, fp64 %.3g), %d MISMATCHES"
          % (len(kerns), exact, len(kerns) - exact - bad, worst["fp32"], worst["fp64"], bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
