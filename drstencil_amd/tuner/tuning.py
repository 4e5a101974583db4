#!/usr/bin/env python3
"""tuning.py -- tuning-space enumerator and search driver for the MI355X generator.

Counterpart of the reference's per-stencil benchmarks/<stencil>/tuning.py
(FilterParams 13-37, cfgToCommandLine 40-58, cfgToString 61-78, getMetrics 84-100,
searchSpace 102-142 in benchmarks/3d7pt_star/tuning.py; the 2D variant in
benchmarks/2d5pt_star/tuning.py:13-164).  Same structure -- filter -> shuffled exhaustive
sweep -> per-config generate/compile/measure -> append every improvement to duration.log
-- but the space is HIP/CDNA4 shaped and the objective comes from HIP events and
rocprofv3 instead of Nsight Compute:

  space vector = (step, dist, (bx, by), sn, unroll, blockMergeX, mx, blockMergeY, my,
                  mergeForward, prefetch, xrim, temporal, xcd, streaming)
  * bx in {16,32,64,128,256}: lanes along x; mx in {1,2,4} points per lane (16-byte
    accesses at mx=4 fp32 / 2 fp64); by*my rows per tile; sn planes per stream block
  * prefetch = planes of software prefetch in flight (0 off, True == 1; `p` / `p3` in the name)
  * LDS budget 160 KiB per CU (the reference caps at 32 KiB of A100 shared memory)
  * `duration` = average kernel duration in ns from HIP events over the reference's timed
    loop (10 warm-up launches first, codegen.hpp:575-584), i.e. what `ncu ... Duration` was
    for the reference; `rocprof_metrics()` adds FETCH_SIZE/WRITE_SIZE for chosen configs.

The config-name scheme (`fu2d1bx16y8sn8u4cmx1cmy1mf5[p]`) and the command-line mapping are
the reference's, with suffixes for the additive options, so logs stay comparable.
"""
import argparse
import datetime
import itertools
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

maxThreadsPerBlockLg2 = 10   # 1024 lanes per workgroup
maxLdsPerBlock = 160 * 1024  # bytes of LDS a single workgroup may use on gfx950
order = 1                    # stencil order (set per stencil by main())
ndim = 3
elem_bytes = 4


def FilterParams(spaceVector):
    step, dist, blockSize, sn, s_unroll, blockMergeX, mergeFactorX, blockMergeY, mergeFactorY, m_threshold, prefetch, xrim, temporal, xcd, streaming = spaceVector
    halo = step * order
    tx = mergeFactorX * blockSize[0]
    ty = 1 if streaming else mergeFactorY * blockSize[1]
    # LDS: 2 planes (3 for an odd number of on-chip stages) x (rows + halo pads) x (row + x halo)
    stage_halo = order if temporal else halo
    slots = 3 if (temporal and step % 2 == 1 and step > 1) else 2
    ldsUsage = slots * (ty + 2 * stage_halo) * (tx + 8) * elem_bytes
    if temporal and step == 1:
        return False
    # the scatter schedule keeps (Range-1) partial-sum planes + one window per stage in VGPRs
    stages = step if temporal else 1
    vgprEstimate = stages * (3 * mergeFactorX * mergeFactorY + 2 * (mergeFactorX + mergeFactorY)) * (1 if temporal or step == 1 else step * step)
    if vgprEstimate > 200:
        return False
    if ldsUsage > maxLdsPerBlock:
        return False
    # dist too big or too small (reference rule; dist > halo is wrong in the reference)
    if dist > step * order or dist < (step - 1) * order:
        return False
    # the tile must keep rows after removing the y halo, and x halo must fit the tile
    # (the generator keeps the reference's "Invalid configuration!" rule: 2*Halo >= tile extent is rejected)
    if (not streaming and ty - 2 * halo < 1) or tx <= 2 * halo:
        return False
    # a wavefront is 64 lanes: idle lanes of the last wave may cost at most 6 % of the workgroup
    threads = blockSize[0] * max(blockSize[1], 1)
    if threads > 1024 or (-threads) % 64 > 0.06 * threads:
        return False
    # one row segment per wavefront instruction wants >= 128 contiguous bytes
    if tx * elem_bytes < 128:
        return False
    # duplicates: block merging with factor 1 == cyclic merging with factor 1
    if blockMergeX and mergeFactorX == 1:
        return False
    if blockMergeY and mergeFactorY == 1:
        return False
    # odd lane counts only pay with temporal blocking (they make the owned width a power of two)
    if blockSize[0] % 16 != 0 and not temporal:
        return False
    # cyclic x lays points out like block x on CDNA4 (plan.hpp): keep one of them
    if not blockMergeX and mergeFactorX > 1:
        return False
    return True


def cfgToCommandLine(spaceVector):
    step, dist, blockSize, sn, s_unroll, blockMergeX, mergeFactorX, blockMergeY, mergeFactorY, m_threshold, prefetch, xrim, temporal, xcd, streaming = spaceVector
    cmd = " --bx {0} --by {1} --sn {2} --stream-unroll {3}".format(blockSize[0], blockSize[1], sn, s_unroll)
    cmd += " --step {0} --dist {1}".format(step, dist)
    if blockMergeX:
        cmd += " --block-merge-x {0}".format(mergeFactorX)
    else:
        cmd += " --cyclic-merge-x {0}".format(mergeFactorX)
    if blockMergeY:
        cmd += " --block-merge-y {0}".format(mergeFactorY)
    else:
        cmd += " --cyclic-merge-y {0}".format(mergeFactorY)
    cmd += " --merge-forward {0}".format(m_threshold)
    if prefetch:
        cmd += " --prefetch --prefetch-depth {0}".format(int(prefetch))   # planes in flight (True == 1)
    cmd += " --xrim {0} --xcd-remap {1}".format(xrim, xcd)
    if temporal:
        cmd += " --temporal 1"
    if streaming:
        cmd += " --streaming"
    return cmd


def cfgToString(spaceVector):
    step, dist, blockSize, sn, s_unroll, blockMergeX, mergeFactorX, blockMergeY, mergeFactorY, m_threshold, prefetch, xrim, temporal, xcd, streaming = spaceVector
    cmd = "fu{0}d{1}bx{2}y{3}sn{4}u{5}".format(step, dist, blockSize[0], blockSize[1], sn, s_unroll)
    cmd += ("bmx{0}" if blockMergeX else "cmx{0}").format(mergeFactorX)
    cmd += ("bmy{0}" if blockMergeY else "cmy{0}").format(mergeFactorY)
    cmd += "mf{0}".format(m_threshold)
    if prefetch:
        cmd += "p" if int(prefetch) == 1 else "p{0}".format(int(prefetch))
    cmd += "x" + xrim[0] + "m" + str(xcd) + ("t" if temporal else "") + ("s" if streaming else "")
    return cmd


def enumerate_space(steps=(1,), full=False):
    """The sweep space.  Lane counts include non-powers of two: with temporal blocking a tile owns
    mx*bx - 2*roundup((step-1)*order, mx) columns, so e.g. bx = 66 (264 columns, 256 owned) tiles a
    1024-wide grid exactly where bx = 64 would need a fifth tile."""
    vec = 16 // elem_bytes
    if ndim == 3:
        bxs = [16, 32, 34, 64, 66, 128, 130, 256]
        bys = [1, 2, 4, 7, 8, 15, 16, 30]
        sns = [4, 8, 16, 32, 64] if not full else [2, 4, 8, 12, 16, 24, 32, 48, 64, 128]
        mys = [1, 2, 4, 8]
    else:
        bxs = [64, 66, 128, 130, 256]
        bys = [1, 2, 4, 8, 15, 16]
        sns = [8, 16, 32, 64, 128]
        mys = [1, 2, 4, 8, 16]
    blockSizes = [(bx, by) for bx in bxs for by in bys if bx * by <= 2 ** maxThreadsPerBlockLg2]
    space = itertools.product(
        list(steps),
        [0],                       # dist: filled per step below
        blockSizes, sns,
        [4, 8] if full else [4],   # stream unroll
        [True], [vec] if not full else [vec // 2, vec],
        [True], mys,
        [5],
        [0, 1, 3] if ndim == 3 else [0, 1],   # software prefetch: planes in flight (0 = off)
        ["dpp"] if not full else ["lds", "dpp"],
        [False, True],             # temporal blocking (only meaningful for step > 1)
        [0, 2] if ndim == 3 else [0],
        [False] if ndim == 3 else [False, True],   # 2D: one-shot tile kernel or --streaming (rows streamed)
    )
    out = []
    for v in space:
        v = list(v)
        v[1] = v[0] * order
        if v[8] == 1:
            v[7] = False           # merge factor 1: cyclic == block, keep the reference's spelling
        if ndim == 2 and not v[14]:
            if v[3] != sns[0] or v[10]:
                continue           # the one-shot tile kernel has no stream blocks and nothing to prefetch
        if ndim == 2 and v[14]:
            if v[2][1] != 1 or v[8] != 1:
                continue           # 2D --streaming ignores by / y merging (codegen_2d.hpp:125)
        v = tuple(v)
        if FilterParams(v):
            out.append(v)
    return out


def getElapsedTime(start, end):
    return (end - start).seconds + (end - start).microseconds / 1e6


def measure(kern, torch, A, B, iterations, warmup=10):
    """HIP-event duration of one launch in ns, reference protocol (10 warm-ups, timed loop)."""
    n, ms = kern.run_timed(A.data_ptr(), B.data_ptr(), iterations=iterations, warmup=warmup, stream=torch.cuda.current_stream().cuda_stream)
    return ms * 1e6 / max(n, 1)


def _build(job):
    import drstencil_amd as drs
    name, args = job
    try:
        k = drs.Kernel(args)
        return name, True, k.path
    except Exception as e:
        return name, False, str(e)[-400:]


def _noop(_):
    return os.getpid()


def searchSpace(stc, is3d, dtype, configs, outdir, iterations=8, budget_s=None, jobs=16, extra_opts=()):
    """Sweep `configs` (space vectors or raw option strings).

    Generation + compilation (hipcc) runs in worker processes that are started from a fork SERVER before
    this process touches the GPU -- a process that has initialised HIP must not fork/exec compilers, the
    clean workers may at any time -- and overlaps with the timing of the configurations already built.
    Improvements go to duration.log (seconds-since-start, best ns, name), every result to results.jsonl;
    a wall-clock budget (the reference's 2D tuner stops after 3600 s) ends the sweep cleanly."""
    import multiprocessing
    from concurrent.futures import ProcessPoolExecutor
    os.makedirs(outdir, exist_ok=True)
    startTime = datetime.datetime.now()
    named = []
    for c in configs:
        if isinstance(c, str):
            named.append((c.strip().replace(" ", "").replace("--", "_"), c.split()))
        else:
            named.append((cfgToString(c), cfgToCommandLine(c).split()))
    base = (["--3d"] if is3d else []) + ["--dtype", dtype] + list(extra_opts)
    jobsl = [(n, base + a + [stc]) for n, a in named]
    argmap = dict(jobsl)
    t_start = time.time()
    pool = ProcessPoolExecutor(max_workers=jobs, mp_context=multiprocessing.get_context("forkserver"))
    list(pool.map(_noop, range(jobs)))             # start the fork server and the workers now, before HIP is up
    futures = [pool.submit(_build, j) for j in jobsl]

    import drstencil_amd as drs
    torch = None
    A = B = None
    best = 1e18
    results = []
    esz = 4 if dtype == "fp32" else 8
    nfail = 0
    for cnt, fut in enumerate(futures, 1):
        name, ok, info = fut.result()
        if not ok:
            nfail += 1
            print("{0}/{1}: {2} BUILD FAILED {3}".format(cnt, len(jobsl), name, info.splitlines()[-1] if info else ""), flush=True)
            continue
        kern = drs.Kernel(argmap[name])                # cache hit: no compiler runs in this process
        if torch is None:
            import torch
            spec = drs.Spec(stc, 3 if is3d else 2, 1)
            L, M, N = spec.dims
            shape = (L, M, N) if is3d else (M, N)
            tdt = torch.float32 if dtype == "fp32" else torch.float64
            A = torch.rand(shape, dtype=tdt, device="cuda")
            B = torch.zeros_like(A)
        dur = measure(kern, torch, A, B, iterations)
        gbs = 2.0 * esz * A.numel() / dur
        gst = kern.updates_per_launch() / dur
        rec = dict(name=name, args=" ".join(argmap[name][:-1]), duration_ns=dur, GBps=gbs, frac=gbs / 8000.0, GStencil=gst,
                   lds=kern.info["lds_bytes"], threads=kern.info["threads"], step=kern.info["step"])
        results.append(rec)
        with open(os.path.join(outdir, "results.jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")
        print("{0}/{1}: {2}  {3:.0f} ns  {4:.0f} GB/s ({5:.1f}%)  {6:.1f} GStencil/s".format(
            cnt, len(jobsl), name, dur, gbs, gbs / 80.0, gst), flush=True)
        if dur < best:
            best = dur
            with open(os.path.join(outdir, "duration.log"), "a") as f:
                f.write(str((datetime.datetime.now() - startTime).seconds) + " s, " + str(int(best)) + ", " + name + "\n")
        if budget_s and time.time() - t_start > budget_s:
            print("time budget reached after {0} of {1} configurations".format(cnt, len(jobsl)), flush=True)
            for f in futures[cnt:]:
                f.cancel()
            break
    pool.shutdown(wait=False, cancel_futures=True)
    with open(os.path.join(outdir, "duration.log"), "a") as f:
        f.write(str((datetime.datetime.now() - startTime).seconds) + " s, " + str(int(best)) + "\n")
    # the reference minimises Duration for a fixed step; across steps the objective is updates per second
    results.sort(key=lambda r: -r["GStencil"])
    return results


def main():
    global order, ndim, elem_bytes
    ap = argparse.ArgumentParser(description="tuning-space search for one stencil on the local MI355X")
    ap.add_argument("stc")
    ap.add_argument("--3d", dest="is3d", action="store_true")
    ap.add_argument("--dtype", default="fp32")
    ap.add_argument("--order", type=int, default=1)
    ap.add_argument("--steps", default="1")
    ap.add_argument("--max-configs", type=int, default=0, help="random subset size (0 = all)")
    ap.add_argument("--budget", type=float, default=0, help="wall-clock budget in seconds (reference 2D: 3600)")
    ap.add_argument("--out", default="tuning_out")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--list", action="store_true", help="only print the space size and the first configs")
    ap.add_argument("--configs-file", default=None, help="file with one raw option string per line instead of the space")
    a = ap.parse_args()
    order, ndim, elem_bytes = a.order, (3 if a.is3d else 2), (4 if a.dtype == "fp32" else 8)
    if a.configs_file:
        paras = [l.strip() for l in open(a.configs_file) if l.strip() and not l.startswith("#")]
    else:
        paras = enumerate_space(tuple(int(s) for s in a.steps.split(",")))
        random.seed(a.seed)
        random.shuffle(paras)
        if a.max_configs:
            paras = paras[:a.max_configs]
    if a.list:
        print(len(paras), "configurations")
        for p in paras[:5]:
            print(cfgToString(p) if not isinstance(p, str) else p, "|", cfgToCommandLine(p) if not isinstance(p, str) else "")
        return
    res = searchSpace(os.path.abspath(a.stc), a.is3d, a.dtype, paras, a.out, budget_s=a.budget or None)
    print("best:")
    for r in res[:10]:
        print("  {name}  {duration_ns:.0f} ns  {GBps:.0f} GB/s  {GStencil:.1f} GStencil/s".format(**r))


if __name__ == "__main__":
    main()
