#!/usr/bin/env python3
"""Randomised sweep of the multi-GPU slab path on ONE GPU: random stencil shapes (tests/fuzz_shapes.py: one-sided, sparse,
dense, orders 1-2), random world sizes 2-8, both exchange modes, a few option sets per shape, with and without separate
"alone" kernels for the launches that run without an exchange beside them (bench.slab_alone_options' role).  Every rank of the world
runs in turn on this GPU through the product SlabRun + HipSweep code (slab views, boundary/interior launches, the
--pair-launch boundary kernel of middle ranks, ghost planes, streams and events; the transport is an in-process mailbox) and the
owned planes of every rank must equal the single-domain run of the same kernel bit for bit -- the property
tests/test_gpu_parity.py::test_slab_decomposition_on_one_gpu checks for the hand-written stencils.
usage: fuzz_slabs.py <shapes> <seed>.  FUZZ_BUILD_ONLY=1 fills the kernel cache on a box without a GPU."""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import drstencil_amd as drs
from drstencil_amd.multigpu import HipSweep, SlabPlan, SlabRun
from helpers import write_stc
from concurrent.futures import ProcessPoolExecutor
import fuzz_shapes as fs

CACHE = os.environ.get("DRS_KCACHE", os.path.join(ROOT, "drstencil_amd", "_kcache"))
OPTION_SETS_3D = [["--sn", "8"], ["--sn", "16", "--prefetch"], ["--sn", "8", "--schedule", "window"], ["--sn", "12", "--dist", "{d}"], ["--sn", "8", "--step", "2"],
                  ["--sn", "8", "--step", "2", "--dist", "{d2}", "--prefetch"], ["--sn", "8", "--stage", "dma"], ["--sn", "16", "--step", "2", "--schedule", "scatter", "--prefetch", "--prefetch-depth", "2"],
                  # round 4: the XCD unit / chunk maps and the strided x merge in slab-view kernels
                  ["--sn", "8", "--step", "2", "--xcd-remap", "4"], ["--sn", "8", "--xcd-remap", "5", "--xcd-chunk", "3", "--prefetch"],
                  ["--sn", "8", "--cyclic-merge-x", "2", "--bx", "32", "--by", "4", "--block-merge-y", "2"]]
OPTION_SETS_2D = [[], ["--streaming", "--sn", "16"], ["--streaming", "--sn", "32", "--prefetch", "--dist", "{d}"], ["--step", "2"], ["--streaming", "--sn", "16", "--step", "2", "--schedule", "scatter"]]


def make_jobs(nshapes, seed):
    """(stc, ndim, step, halo, dims, dtype, options, world, every, alone options or None) per job."""
    rnd = random.Random(seed)
    out = os.path.join(ROOT, "gpurun_out", "fuzz_slabs"); os.makedirs(out, exist_ok=True)
    jobs = []
    for s in range(nshapes):
        ndim = rnd.choice([2, 3, 3])
        h = rnd.choice([1, 1, 2])
        pts, _mixed = fs.random_shape(rnd, ndim, h)
        d1, d2 = fs.legal_dists(pts, 1), fs.legal_dists(pts, 2)
        if not d1:
            continue
        # the slab dimension is long enough for a handful of ranks of at least 2 * every * step * halo planes each (thinner: refused)
        dims = (rnd.randint(44, 170), rnd.randint(20, 40), rnd.randint(70, 200)) if ndim == 3 else (1, rnd.randint(60, 300), rnd.randint(150, 500))
        stc = os.path.join(out, "slab%d_s%d_%dd_o%d.stc" % (seed, s, ndim, h))
        write_stc(stc, ndim, dims, rnd.choice([2, 4, 6]), pts)
        for opt in rnd.sample(OPTION_SETS_3D if ndim == 3 else OPTION_SETS_2D, 2):
            opt = [x.format(d=rnd.choice(d1), d2=rnd.choice(d2)) for x in opt]
            step = int(opt[opt.index("--step") + 1]) if "--step" in opt else 1
            if len(set(p[:-1] for p in pts)) ** step > 300:
                continue
            dtype = rnd.choice(["fp32", "fp64"])
            opts = (["--3d"] if ndim == 3 else []) + ["--dtype", dtype] + opt
            alone = None
            if "--sn" in opts and rnd.random() < 0.4:      # whole-slab launches of an every = 2 pair on their own kernel: one stream block per tile
                alone = list(opts); alone[alone.index("--sn") + 1] = "400"
            jobs.append((stc, ndim, step, h * step, dims, dtype, opts, rnd.randint(2, 8), rnd.choice([1, 2]), alone))
    return jobs


def build(job):
    stc, ndim, step, halo, dims, dtype, opts, world, every, alone = job
    try:
        drs.Kernel(opts + [stc])
        sw = HipSweep(stc, opts, CACHE, alone_opts=alone)
        cut = dims[0] if ndim == 3 else dims[1]
        for r in range(world):
            sw.prebuild(SlabPlan(cut, halo, world, r, every))
        return None
    except Exception as e:
        return "%s: %s" % (" ".join(opts), " ".join(str(e).split())[:200])


class _Hub:
    def __init__(self):
        self.mail, self.pending = {}, []

    def deliver(self):
        still = []
        for key, t in self.pending:
            q = self.mail.get(key)
            if q:
                t.copy_(q.pop(0))
            else:
                still.append((key, t))
        self.pending = still


class _Mailbox:
    """torch.distributed's point-to-point surface as SlabRun uses it, delivered in process."""
    isend, irecv = "isend", "irecv"

    class P2POp:
        def __init__(self, op, tensor, peer):
            self.op, self.tensor, self.peer = op, tensor, peer

    class _Work:
        def wait(self):
            return True

    def __init__(self, hub, rank, torch):
        self.hub, self.rank, self.torch = hub, rank, torch

    def batch_isend_irecv(self, ops):
        for o in ops:
            if o.op == "isend":
                self.hub.mail.setdefault((self.rank, o.peer), []).append(o.tensor.clone())
            else:
                self.hub.pending.append(((o.peer, self.rank), o.tensor))
        self.torch.cuda.current_stream().synchronize()
        self.hub.deliver()
        return [self._Work() for _ in ops]


def check(job, torch):
    stc, ndim, step, halo, dims, dtype, opts, world, every, alone = job
    full = drs.Kernel(opts + [stc])
    L, M, N = dims
    shape = (L, M, N) if ndim == 3 else (M, N)
    tdt = torch.float32 if dtype == "fp32" else torch.float64
    g = torch.Generator(device="cuda").manual_seed(11)
    A0 = torch.rand(shape, dtype=tdt, device="cuda", generator=g)
    A, B = A0.clone(), torch.zeros_like(A0)
    iterations = full.info["iterations"]
    full.run(A.data_ptr(), B.data_ptr())
    torch.cuda.synchronize()
    hub = _Hub()
    dev = torch.device("cuda", 0)
    sweep = HipSweep(stc, opts, CACHE, alone_opts=alone)
    runs = [SlabRun(torch, _Mailbox(hub, r, torch), shape, halo, step, iterations, r, world, sweep, dev, tdt, every=every) for r in range(world)]
    for r in runs:
        r.load_global(lambda lo, hi: A0[lo:hi])
    t = 0
    while t < iterations:
        for src, dst in (("A", "B"), ("B", "A")):
            for r in runs:
                if every == 2 and src == "A":
                    r.launch_local(r.A, r.B)
                else:
                    r.launch(getattr(r, src), getattr(r, dst))
            torch.cuda.synchronize()
            hub.deliver()
            if hub.pending:
                return False
        t += 2 * step
    for r in runs:
        p = r.plan
        if not (torch.equal(r.owned(r.A), A[p.z0:p.z1]) and torch.equal(r.owned(r.B), B[p.z0:p.z1])):
            return False
    return True


def main():
    jobs = make_jobs(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    t0 = time.time()
    with ProcessPoolExecutor(max_workers=int(os.environ.get("FUZZ_JOBS", "16"))) as ex:
        errs = list(ex.map(build, jobs))
    ok = [j for j, e in zip(jobs, errs) if e is None]
    rej = [e for e in errs if e is not None]
    why = {}
    for e in rej:
        msg = e.split(": ", 1)[1] if ": " in e else e
        key = "register spills" if "spills" in msg else " ".join(msg.split()[:10])
        why[key] = why.get(key, 0) + 1
    print("%d slab runs: kernels of %d built in %.0f s; %d rejected by the generator or refused by the runtime" % (len(jobs), len(ok), time.time() - t0, len(rej)), flush=True)
    for k, n in sorted(why.items(), key=lambda kv: -kv[1])[:8]:
        print("  %5d x %s" % (n, k))
    if os.environ.get("FUZZ_BUILD_ONLY"):
        return
    import torch
    bad = 0
    for cnt, job in enumerate(ok, 1):
        if cnt % 25 == 0:
            print("... %d / %d slab runs checked, %d mismatches" % (cnt, len(ok), bad), flush=True)
        try:
            good = check(job, torch)
        except Exception as e:
            good = False
            print("ERROR", " ".join(str(e).split())[:300], flush=True)
        if not good:
            bad += 1
            print("MISMATCH world %d every %d%s" % (job[7], job[8], " alone-kernels" if job[9] else ""), " ".join(job[6]), os.path.basename(job[0]), flush=True)
    print("%d slab runs (worlds 2-8, both exchange modes) checked against the single-domain kernel: %d MISMATCHES" % (len(ok), bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
