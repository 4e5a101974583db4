#!/usr/bin/env python3
"""Exploration harness (GPU box): times generated kernels for ad-hoc (dims, options) pairs
and a device-to-device copy as the practical HBM ceiling.  Builds everything before HIP is
initialised.  Usage: explore.py <experiment-file> <outdir>; experiment lines:
    <name> | <ndim> | <L> <M> <N> | <stencil: star3|star2|box25> | <dtype> | <options...>"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import drstencil_amd as drs
from helpers import write_stc

STAR3 = [(0, 0, 0, 0.3), (1, 0, 0, 0.2), (-1, 0, 0, 0.2), (0, 1, 0, 0.2), (0, -1, 0, 0.2), (0, 0, 1, 0.2), (0, 0, -1, 0.2)]
STAR2 = [(0, 0, 0.3), (0, 1, 0.2), (1, 0, 0.2), (0, -1, 0.2), (-1, 0, 0.2)]
BOX25 = [(0, 0, 0.3), (1, 0, 0.2), (0, 1, 0.2), (-1, 0, 0.2), (0, -1, 0.2), (1, 1, 0.1), (1, -1, 0.1), (-1, 1, 0.1), (-1, -1, 0.1),
         (2, 0, 0.1), (0, 2, 0.1), (-2, 0, 0.1), (0, -2, 0.1), (1, 2, 0.05), (-1, 2, 0.05), (1, -2, 0.05), (-1, -2, 0.05),
         (-2, 1, 0.05), (2, -1, 0.05), (-2, -1, 0.05), (2, 1, 0.05), (2, 2, 0.02), (2, -2, 0.02), (-2, 2, 0.02), (-2, -2, 0.02)]
PTS = dict(star3=STAR3, star2=STAR2, box25=BOX25)


def _build(job):
    try:
        k = drs.Kernel(job)
        return True, k.path
    except Exception as e:
        return False, str(e)[-300:]


def main():
    exp, outdir = sys.argv[1], sys.argv[2]
    os.makedirs(outdir, exist_ok=True)
    jobs = []
    for line in open(exp):
        line = line.strip()
        if not line or line.startswith("#"):
            continue
        name, ndim, dims, pts, dtype, opts = [s.strip() for s in line.split("|")]
        ndim = int(ndim)
        d = [int(x) for x in dims.split()]
        if ndim == 2:
            d = [1] + d
        stc = os.path.join(outdir, "x%s_%s.stc" % (pts.lstrip("@"), "x".join(str(x) for x in d)))
        if not os.path.exists(stc):
            if pts.startswith("@"):       # the point list of a shipped spec: benchmarks/<name>/<name>.stc
                body = open(os.path.join(ROOT, "benchmarks", pts[1:], pts[1:] + ".stc")).read().split("stencil", 1)[1]
                table = [tuple(float(x) if "." in x else int(x) for x in ln.split()) for ln in body.strip().splitlines() if ln.strip()]
            else:
                table = PTS[pts]
            write_stc(stc, ndim, tuple(d), 4, table)
        args = (["--3d"] if ndim == 3 else []) + ["--dtype", dtype] + opts.split() + [stc]
        jobs.append((name, ndim, d, dtype, args))
    from concurrent.futures import ProcessPoolExecutor
    t0 = time.time()
    with ProcessPoolExecutor(max_workers=min(16, os.cpu_count())) as ex:
        res = list(ex.map(_build, [j[4] for j in jobs]))
    print("built in %.0f s" % (time.time() - t0), flush=True)
    if os.environ.get("EXPLORE_BUILD_ONLY"):      # fill the kernel cache on a box without a GPU (the cache travels with the tree)
        for j, (ok, info) in zip(jobs, res):
            if not ok:
                print(j[0], "BUILD FAILED", info.splitlines()[-1] if info else "", flush=True)
            else:       # what the compiler allocated (cache hit: the stored report), beside what the generator names
                k = drs.Kernel(j[4])
                r, i = k.resources, k.info
                print("%-32s vgprs %3s agprs %3s scratch %s occ %s lds %6d  named %3d  %d lanes, grid %d, %s" % (j[0], r.get("vgprs"), r.get("agprs"), r.get("scratch_bytes_per_lane"),
                      r.get("occupancy_waves_per_simd"), i["lds_bytes"], i["reg_demand"], i["threads"], i["grid"], i.get("arithmetic")), flush=True)
        return
    kerns = []
    for j, (ok, info) in zip(jobs, res):
        if not ok:
            print(j[0], "BUILD FAILED", info.splitlines()[-1] if info else "", flush=True)
            continue
        kerns.append((j, drs.Kernel(j[4])))
    import torch
    bufs = {}
    out = open(os.path.join(outdir, "explore.jsonl"), "a")
    # copy ceiling
    for nbytes in (1 << 30, 4 << 30):
        a = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda").uniform_()
        b = torch.empty_like(a)
        for _ in range(3):
            b.copy_(a)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("copy %d MiB: %.3f ms  %.0f GB/s (read+write)" % (nbytes >> 20, ms, 2 * nbytes / ms / 1e6), flush=True)
        out.write(json.dumps(dict(name="copy_%dMiB" % (nbytes >> 20), ms=ms, GBps=2 * nbytes / ms / 1e6)) + "\n")
        del a, b
    if os.environ.get("EXPLORE_VERIFY"):
        # one launch of every kernel against its own gold kernel (bit for bit; temporal pipelines: max relative difference printed)
        for (name, ndim, d, dtype, args), k in kerns:
            if name.startswith("gold") or not k.info.get("valid", 1):
                continue
            tdt = torch.float32 if dtype == "fp32" else torch.float64
            shape = tuple(d) if ndim == 3 else tuple(d[1:])
            g = torch.Generator(device="cuda").manual_seed(5)
            A = torch.rand(shape, dtype=tdt, device="cuda", generator=g)
            B = torch.zeros_like(A); G = torch.zeros_like(A)
            k.launch(A.data_ptr(), B.data_ptr()); k.launch_gold(A.data_ptr(), G.data_ptr())
            torch.cuda.synchronize()
            if torch.equal(B, G):
                print("verify %-28s == gold kernel, bit for bit (%s, %s VGPRs, %s scratch)" % (name, k.info.get("arithmetic"), k.resources.get("vgprs"), k.resources.get("scratch_bytes_per_lane")), flush=True)
            else:
                h = k.info["halo"]
                inner = tuple(slice(h, n - h) for n in shape)
                rel = float(((B[inner] - G[inner]).abs() / G[inner].abs().clamp_min(1e-30)).max())
                print("verify %-28s DIFFERS from the gold kernel: max rel %.3g (%s)" % (name, rel, k.info.get("arithmetic")), flush=True)
            del A, B, G
        torch.cuda.empty_cache()
    rounds = int(os.environ.get("EXPLORE_ROUNDS", "1"))
    agg = {}
    order_names = []
    for rnd in range(rounds):
      for (name, ndim, d, dtype, args), k in kerns:
          key = (tuple(d), dtype)
          if key not in bufs:
              # Buffers live for the whole run (288 GB of HBM): until this round's fix they were re-allocated whenever the grid changed, i.e.
              # once per round and grid, and the first ~140 ms of launches on a FRESH pair of 4 GiB arrays run the fused step-3 kernels
              # 15-20 % slower (scripts/archive/probe_cold.py) -- what earlier logs read as a configuration or a device being slow was the
              # position of the line in the file.  Both arrays sit in one arena at the first kernel's recommended skew, and every
              # pair is warmed for EXPLORE_WARM_S (default 0.4 s) before its first timing.
              tdt = torch.float32 if dtype == "fp32" else torch.float64
              if os.environ.get("EXPLORE_SEPARATE"):
                  shape = tuple(d) if ndim == 3 else tuple(d[1:])
                  pair = (torch.rand(shape, dtype=tdt, device="cuda"), torch.zeros(shape, dtype=tdt, device="cuda"), None)
              else:
                  pair = k.alloc_pair(torch, "cuda", dtype=tdt, skew=int(os.environ["EXPLORE_SKEW_MIB"]) << 20 if os.environ.get("EXPLORE_SKEW_MIB") else None)
                  pair[0].uniform_(); pair[1].zero_()
              bufs[key] = pair
              tw = time.time()
              while time.time() - tw < float(os.environ.get("EXPLORE_WARM_S", "0.4")) and not name.startswith("gold"):
                  for _ in range(4):
                      k.launch(pair[0].data_ptr(), pair[1].data_ptr()); k.launch(pair[1].data_ptr(), pair[0].data_ptr())
                  torch.cuda.synchronize()
          A, B = bufs[key][:2]
          # finite data in every timed loop: the test stencils' coefficients sum to more than 1 (star3: 1.5), an array started from U[0, 1)
          # is all inf after ~218 time steps, and on inf / NaN operands the VALU-dense kernels run 15-20 % FASTER (clocks; probe_cold2.py)
          if not os.environ.get("EXPLORE_NO_RESEED"):
              A.uniform_(); B.zero_()
          if name.startswith("gold"):
              for _ in range(2):
                  k.launch_gold(A.data_ptr(), B.data_ptr())
              torch.cuda.synchronize()
              e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
              e0.record()
              for _ in range(4):
                  k.launch_gold(A.data_ptr(), B.data_ptr())
              e1.record()
              torch.cuda.synchronize()
              n, ms = 4, e0.elapsed_time(e1)
          else:
              hz = k.info.get("tolerance_horizon_iterations", -1)
              if k.info.get("arithmetic") == "reassociated" and not k.info.get("temporal_forced") and 0 < hz < 8 * k.info["step"]:
                  n, ms = 0, 0.0        # a temporal pipeline runs up to its tolerance horizon only: several short loops
                  for rep in range(4):
                      n_, ms_ = k.run_timed(A.data_ptr(), B.data_ptr(), iterations=hz, warmup=4 if rep == 0 else 0, stream=torch.cuda.current_stream().cuda_stream)
                      n, ms = n + n_, ms + ms_
              else:
                  n, ms = k.run_timed(A.data_ptr(), B.data_ptr(), iterations=8 * k.info["step"], warmup=4, stream=torch.cuda.current_stream().cuda_stream)
          dur = ms / n
          gbs = k.bytes_per_launch() / dur / 1e6
          gst = k.updates_per_launch() / dur / 1e6
          rec = dict(name=name, dims=d, dtype=dtype, args=" ".join(args[:-1]), ms=dur, GBps=gbs, frac=gbs / 8000, GStencil=gst, lds=k.info["lds_bytes"], threads=k.info["threads"], grid=k.info["grid"])
          out.write(json.dumps(rec) + "\n")
          out.flush()
          if rounds == 1:
              print("%-28s %8.3f ms %6.0f GB/s (%4.1f%%) %7.1f GSt  lds=%6d grid=%d" % (name, dur, gbs, gbs / 80, gst, k.info["lds_bytes"], k.info["grid"]), flush=True)
          if name not in agg:
              order_names.append(name)
          agg.setdefault(name, []).append((dur, gbs, gst))
    if rounds > 1:
        import statistics
        print("median of %d interleaved rounds:" % rounds)
        for name in order_names:
            v = agg[name]
            dur = statistics.median(x[0] for x in v); gbs = statistics.median(x[1] for x in v); gst = statistics.median(x[2] for x in v)
            print("%-28s %8.3f ms %6.0f GB/s (%4.1f%%) %7.1f GSt   [min %.3f max %.3f ms]" % (name, dur, gbs, gbs / 80, gst, min(x[0] for x in v), max(x[0] for x in v)), flush=True)


if __name__ == "__main__":
    main()
