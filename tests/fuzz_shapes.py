#!/usr/bin/env python3
"""Randomised parity sweep over stencil SHAPES: random point sets inside the order cube (sparse to dense, asymmetric, with or
without the centre, duplicate offsets -- the last one wins, drstencil.hpp:52-78 --, mixed-sign coefficients), 2D and 3D,
orders 1-3, on small ragged grids, each with a few random configurations from the tuner's space (all schedules, every legal
and some illegal --dist) and compared with the CPU oracle like tests/fuzz_parity.py: bit for bit for single-pass kernels,
within the dtype's bar for temporal pipelines.  The hand-written test stencils are stars, boxes and crosses; this sweep is
for everything else the .stc format can say.
usage: fuzz_shapes.py <shapes> <configurations per shape and dtype> <seed>.  FUZZ_BUILD_ONLY=1 fills the kernel cache on a
box without a GPU (the .stc files are regenerated from the seed on either side)."""
import itertools, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tests/ may use the oracle as the checker
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import drstencil_amd as drs
from drstencil_amd.tuner import tuning as t
from helpers import write_stc
from concurrent.futures import ProcessPoolExecutor
import fuzz_parity as fp

BIG = bool(os.environ.get("FUZZ_SHAPES_BIG"))
dropped = [0]         # configurations the tuner's spill model kept from the compiler
MAX_TAPS = int(os.environ.get("FUZZ_MAX_TAPS", "420"))        # fused point count above which a step is not tried (compile time)


def random_shape(rnd, ndim, h):
    """(points, mixed_sign): a random subset of [-h, h]^ndim that contains a point at +h along the outermost dimension (the
    reference takes Halo from there, drstencil.hpp:182-196, and every other offset must stay within it)."""
    cells = list(itertools.product(range(-h, h + 1), repeat=ndim))
    density = rnd.choice([0.12, 0.25, 0.5, 0.8, 1.0])
    chosen = [c for c in cells if rnd.random() < density]
    top = tuple([h] + [rnd.randint(-h, h) if rnd.random() < 0.3 else 0 for _ in range(ndim - 1)])
    if not any(c[0] == h for c in chosen):
        chosen.append(top)
    if rnd.random() < 0.3:      # one-sided along the streamed dimension
        chosen = [c for c in chosen if c[0] >= 0] or [top]
    rnd.shuffle(chosen)
    mixed = rnd.random() < 0.25
    if not mixed and rnd.random() < 0.45:
        # two-decimal coefficients: products of up to three of them have six decimals, so rounding the fused coefficients is a
        # no-op and --temporal 1 really emits on-chip stages (planner.hpp falls back to the fused kernel otherwise)
        unit = max(1, int(90 / len(chosen)))
        pts = [c + (rnd.randint(1, unit) / 100.0,) for c in chosen]
    else:
        w = [rnd.uniform(0.02, 1.0) for _ in chosen]
        scale = rnd.uniform(0.7, 1.0) / sum(w)
        pts = [c + ((-1.0 if mixed and rnd.random() < 0.3 else 1.0) * x * scale,) for c, x in zip(chosen, w)]
    if rnd.random() < 0.2:      # a duplicate offset: the later line replaces the earlier one
        c = rnd.choice(pts)
        pts.append(c[:-1] + (c[-1] * 0.5,))
    return pts, mixed


def legal_dists(pts, step):
    """The --dist values for which the reference finds data to reuse (drstencil.hpp:198-259): some point of the fused
    stencil has another one `dist` behind it along the outermost dimension."""
    offs = {p[:-1] for p in pts}
    fused = {tuple([0] * len(next(iter(offs))))}
    for _ in range(step):
        fused = {tuple(a + b for a, b in zip(f, o)) for f in fused for o in offs}
    span = max(f[0] for f in fused) - min(f[0] for f in fused)
    return [d for d in range(1, span + 1) if any((f[0] - d,) + f[1:] in fused for f in fused)]


def make_jobs(nshapes, per, seed):
    rnd = random.Random(seed)
    out = os.path.join(ROOT, "gpurun_out", "fuzz_shapes"); os.makedirs(out, exist_ok=True)
    jobs = []
    for s in range(nshapes):
        ndim = rnd.choice([2, 3])
        h = rnd.choice([1, 1, 2] if ndim == 3 else [1, 2, 2, 3])
        pts, mixed = random_shape(rnd, ndim, h)
        dims = (rnd.randint(18 + 2 * h, 40), rnd.randint(30, 70), rnd.randint(130, 300)) if ndim == 3 else (1, rnd.randint(90, 260), rnd.randint(200, 600))
        if rnd.random() < 0.25:     # grids barely larger than the ring of a 3-step kernel, narrower than a tile, a vector or a wavefront's row
            lo = 6 * h + 1
            dims = (rnd.randint(lo, lo + 5), rnd.randint(lo, lo + 9), rnd.randint(lo, 48)) if ndim == 3 else (1, rnd.randint(lo, lo + 9), rnd.randint(lo, 70))
        if BIG:                     # BASELINE-sized grids: offsets beyond 2^31 bytes, thousands of tiles; checked against the gold kernel
            dims = (rnd.choice([700, 1024]), rnd.choice([1000, 1024]), rnd.choice([1024, 1100])) if ndim == 3 else (1, rnd.choice([8192, 12000]), rnd.choice([8192, 16384]))
        stc = os.path.join(out, "shape%d_s%d_%dd_o%d.stc" % (seed, s, ndim, h))
        write_stc(stc, ndim, dims, rnd.randint(1, 7), pts)     # iterations: 2 * ceil(iterations / (2 * step)) launches (codegen.hpp:581-584)
        distinct = len(set(p[:-1] for p in pts))
        steps = tuple(st for st in (1, 2, 3) if min((2 * h * st + 1) ** ndim, distinct ** st) <= MAX_TAPS)
        dists = {}
        for dtype in ("fp32", "fp64"):
            t.order, t.ndim, t.elem_bytes = h, ndim, 4 if dtype == "fp32" else 8
            space = t.enumerate_space(steps)
            for v in rnd.sample(space, min(len(space), per)):
                cl = t.cfgToCommandLine(v).split()
                if "--temporal" in cl and mixed:
                    continue                     # a relative bar means nothing where the sum cancels
                r = rnd.random()
                i = cl.index("--dist")
                legal = dists.setdefault(v[0], legal_dists(pts, v[0]))
                if r < 0.2:                      # the reference's range, (step-1)*order .. step*order (tuning.py:20): refused ("No data to reuse") or right
                    cl[i + 1] = str(rnd.randint(max(1, (v[0] - 1) * h), v[0] * h))
                elif r < 0.4 and legal:          # a distance this shape has data to reuse at
                    cl[i + 1] = str(rnd.choice(legal))
                elif r < 0.45:                   # anything: must be refused or right
                    cl[i + 1] = str(rnd.randint(1, 2 * v[0] * h + 1))
                elif r < 0.6:                    # the default, (high - low) / 2
                    del cl[i:i + 2]
                if ndim == 2 and rnd.random() < 0.5:
                    cl.append("--streaming")
                if "--prefetch-depth" in cl:
                    cl[cl.index("--prefetch-depth") + 1] = str(rnd.choice([1, 2, 3, 4]))
                if "--schedule" not in cl and rnd.random() < 0.6:
                    cl[cl.index("--merge-forward") + 1] = str(rnd.choice([0, 2, 3, 100]))
                if rnd.random() < 0.2:
                    cl += ["--uniform-loads", str(rnd.choice([1, 2]))]
                if rnd.random() < 0.2:
                    cl += ["--store-mask", "buffer"]
                if rnd.random() < 0.25 and "--temporal" not in cl and "--cyclic-merge-y" not in cl and (ndim == 3 or "--streaming" in cl):
                    cl += ["--stage", "dma"]
                if rnd.random() < 0.2:
                    cl += ["--defer-stores", "1"]
                if fp.ROUND3:
                    fp.round3_knobs(rnd, cl)
                    fp.round4_knobs(rnd, cl)
                    if "--skew" in cl and ndim == 2 and "--streaming" not in cl:
                        del cl[cl.index("--skew"):cl.index("--skew") + 2]
                args = (["--3d"] if ndim == 3 else []) + ["--dtype", dtype] + cl + [stc]
                if not t.registerFilter(args):   # the tuner's spill model: do not compile what would be refused for scratch
                    dropped[0] += 1
                    continue
                jobs.append((ndim, stc, dtype, args, v[0]))
    return jobs


def check_gold(job, k, torch):
    """FUZZ_SHAPES_BIG: the optimised kernel against the gold kernel of the same emitted source (the oracle would take minutes
    per configuration at these sizes; the gold kernel's arithmetic is pinned to it on the small grids)."""
    ndim, stc, dtype, args, step = job
    info = k.info
    shape = (info["L"], info["M"], info["N"]) if ndim == 3 else (info["M"], info["N"])
    tdt = torch.float32 if dtype == "fp32" else torch.float64
    g = torch.Generator(device="cuda").manual_seed(7)
    A0 = torch.rand(shape, dtype=tdt, device="cuda", generator=g)
    A, B, Ag, Bg = A0.clone(), torch.zeros_like(A0), A0.clone(), torch.zeros_like(A0)
    k.run(A.data_ptr(), B.data_ptr())
    k.run(Ag.data_ptr(), Bg.data_ptr(), gold=True)
    torch.cuda.synchronize()
    if info.get("stages", 1) > 1:
        h = info["halo"]
        inner = tuple(slice(h, n - h) for n in shape)
        rel = max(float(((X - G)[inner].abs() / G[inner].abs().clamp_min(1e-30)).max()) for X, G in ((A, Ag), (B, Bg)))
        ring_ok = True
        for X, G in ((A, Ag), (B, Bg)):
            D = X != G
            D[inner] = False
            ring_ok = ring_ok and not bool(D.any())
        bar = 1e-6 if dtype == "fp32" else 1e-12
        forced = bool(info.get("temporal_forced"))       # only --temporal force kernels may drift (fuzz_parity.check)
        return ("ok" if rel <= bar and ring_ok else "drift" if forced and rel <= 10 * bar and ring_ok else "bad"), True, rel
    return ("ok" if torch.equal(A, Ag) and torch.equal(B, Bg) else "bad"), False, 0.0


def main():
    nshapes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    jobs = make_jobs(nshapes, per, int(sys.argv[3]) if len(sys.argv) > 3 else 1)
    t0 = time.time()
    with ProcessPoolExecutor(max_workers=int(os.environ.get("FUZZ_JOBS", "16"))) as ex:
        errs = list(ex.map(fp.build, jobs, chunksize=4))
    ok_jobs = [j for j, e in zip(jobs, errs) if e is None]
    rejected = [e for e in errs if e is not None]
    why = {}
    for e in rejected:
        msg = e.split(": ", 1)[1] if ": " in e else e
        key = "scratch spills" if "scratch" in msg else " ".join(msg.split()[:9])
        why[key] = why.get(key, 0) + 1
    print("%d shapes, %d configurations (+%d dropped by the tuner's spill model): built %d kernels in %.0f s; %d rejected by the generator or refused by the runtime"
          % (nshapes, len(jobs), dropped[0], len(ok_jobs), time.time() - t0, len(rejected)), flush=True)
    for k, n in sorted(why.items(), key=lambda kv: -kv[1])[:12]:
        print("  %5d x %s" % (n, k))
    if os.environ.get("FUZZ_BUILD_ONLY"):
        return
    kerns = [(j, drs.Kernel(j[3])) for j in ok_jobs]
    import torch
    bad = exact = drift = 0
    worst = {"fp32": 0.0, "fp64": 0.0}
    for cnt, (job, k) in enumerate(kerns, 1):
        if cnt % 100 == 0:
            print("... %d / %d checked, %d mismatches" % (cnt, len(kerns), bad), flush=True)
        status, temporal, rel = check_gold(job, k, torch) if BIG else fp.check(job, k, torch)
        if temporal:
            worst[job[2]] = max(worst[job[2]], rel)
        else:
            exact += status == "ok"
        if status == "drift":
            drift += 1
            print("DRIFT (temporal pipeline beyond the bar, rel %.3g)" % rel, " ".join(job[3][:-1]), os.path.basename(job[1]), flush=True)
        elif status != "ok":
            bad += 1
            print("MISMATCH", " ".join(job[3][:-1]), os.path.basename(job[1]), "rel %.3g" % rel, flush=True)
    print("%d configurations checked: %d single-pass bit-exact, %d temporal within tolerance (worst fp32 %.3g, fp64 %.3g), %d temporal beyond 1e-6 / 1e-12 by rounding drift, %d MISMATCHES"
          % (len(kerns), exact, len(kerns) - exact - bad - drift, worst["fp32"], worst["fp64"], drift, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
