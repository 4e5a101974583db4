#!/usr/bin/env python3
"""bench.py -- the north-star measurement (BASELINE.json): GStencil/s and achieved HBM GB/s
vs the MI355X roofline on 3d7pt_star, fp32.

    python bench.py --gpus N --steps K --warmup W

Workload (config.workload): C4 = 3d7pt_star 1024^3 fp32, `iterations 4` (the configuration
the metric is quoted on; 8 GiB for the two buffers, fits one GPU).  One "step" = the
reference's timed region (codegen.hpp:581-584): the ping-pong loop over the .stc's 4 time
steps = 2*ceil(4/(2*step)) kernel launches.  N > 1: the same 1024^3 grid cut into z slabs,
one process per GPU, RCCL halo exchange overlapped with the interior sweep (strong scaling).

value      = grid-point updates of all ranks / max-over-ranks wall time, in GStencil/s
roofline   = algorithmic bytes (2*sizeof(T) per grid point per launch)
             / average launch duration measured with HIP events on the launch stream
cpu_baseline = the CPU oracle (oracle/, OpenMP) on a bounded z-slab sample of the same workload
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CFG = os.path.join(ROOT, "benchmarks", "configs")
WORKLOADS = {
    "c4": dict(stc=os.path.join(CFG, "c4_3d7pt_star_1024.stc"), ndim=3, dtype="fp32", name="3d7pt_star 1024^3 fp32 (BASELINE C4), iterations 4"),
    "c3": dict(stc=os.path.join(CFG, "c3_3d7pt_star_512.stc"), ndim=3, dtype="fp32", name="3d7pt_star 512^3 fp32 (BASELINE C3), iterations 4"),
    "c2": dict(stc=os.path.join(CFG, "c2_2d5pt_star_8192.stc"), ndim=2, dtype="fp32", name="2d5pt_star 8192^2 fp32 (BASELINE C2), iterations 4"),
    "c5": dict(stc=os.path.join(CFG, "c5_2d25pt_box_16384.stc"), ndim=2, dtype="fp64", name="2d25pt_box 16384^2 fp64 (BASELINE C5), iterations 4"),
}
# tuned generator options per workload (found with drstencil_amd/tuner; logs under profiles/).
# Headline for the 3D workloads: two time steps per launch with the reference's own --step 2 arithmetic
# (algebraically fused 25-point stencil, one pass) -- bit-identical to the oracle.  32 lanes x 4 points per row,
# 16 lane rows x 2 rows (32 x 128 tile, halo fetched by the halo loader lanes), 32-plane stream blocks, three
# planes in flight per lane (software prefetch depth 3: the 200+ VGPRs of the fused kernel leave one workgroup
# per CU, so the bytes in flight have to come from depth -- profiles/r01_exp_r1z*_prefetch_depth.log), one x-y
# band per XCD.  The exhaustive 1780-configuration search
# (profiles/r01_tune_c4_s2_exhaustive.txt) puts the 32x16-lane and 64x8-lane fused kernels and the 66x15-lane
# temporal pipeline within a few per cent of each other; their order changes from device to device.
TUNED = {
    "c4": ["--3d", "--dtype", "fp32", "--step", "2", "--prefetch", "--prefetch-depth", "3", "--bx", "32", "--by", "16", "--block-merge-x", "4", "--block-merge-y", "2", "--sn", "32", "--xcd-remap", "2"],
    "c3": ["--3d", "--dtype", "fp32", "--step", "2", "--prefetch", "--prefetch-depth", "3", "--bx", "32", "--by", "16", "--block-merge-x", "4", "--block-merge-y", "2", "--sn", "32", "--xcd-remap", "2"],
    # 2D one-shot LDS tiles (BASELINE C2 "no temporal blocking (baseline LDS tile)", C5 "wide-halo LDS staging"): best of the
    # exhaustive 2D searches, profiles/r01_tune_c2_exhaustive.txt / r01_tune_c5_exhaustive.txt (0.78 of the HBM peak each)
    "c2": ["--dtype", "fp32", "--bx", "128", "--by", "2", "--block-merge-x", "4", "--block-merge-y", "2", "--xcd-remap", "0"],
    "c5": ["--dtype", "fp64", "--bx", "64", "--by", "4", "--block-merge-x", "2", "--block-merge-y", "4", "--xcd-remap", "0"],
}
# the same workloads with one time step per launch: highest roofline fraction.  Full-row tiles (256 lanes x 4
# points = N), 2 lane rows x 4 rows, 4-plane stream blocks, prefetch: the optimum of the exhaustive 1520-configuration
# search (profiles/r01_tune_c4_s1_exhaustive.txt) -- 80 % of the HBM peak, the chip's measured copy ceiling
STEP1 = {
    "c4": ["--3d", "--dtype", "fp32", "--prefetch", "--bx", "256", "--by", "2", "--block-merge-x", "4", "--block-merge-y", "4", "--sn", "4", "--xcd-remap", "2"],
    "c3": ["--3d", "--dtype", "fp32", "--prefetch", "--bx", "32", "--by", "16", "--block-merge-x", "4", "--block-merge-y", "2", "--sn", "4", "--xcd-remap", "2"],
}
# on-chip temporal blocking (two applications of the one-step stencil per launch; equal to the fused stencil up to
# rounding, 6.8e-7 relative at full size): 66 lanes x 4 = 264 columns own 256, so 4 tiles cover N = 1024 exactly
TEMPORAL2 = {
    "c4": ["--3d", "--dtype", "fp32", "--step", "2", "--temporal", "1", "--prefetch", "--bx", "66", "--by", "15", "--block-merge-x", "4", "--block-merge-y", "2", "--sn", "32", "--xcd-remap", "0"],
    "c3": ["--3d", "--dtype", "fp32", "--step", "2", "--temporal", "1", "--prefetch", "--bx", "66", "--by", "15", "--block-merge-x", "4", "--block-merge-y", "2", "--sn", "16", "--xcd-remap", "0"],
}
# N > 1 (z slabs of C4): the same fused kernel; slabs of 256 planes or fewer get 16-plane stream blocks.  One stream block
# per tile (256 tiles = one workgroup per CU) is the fastest way to sweep a slab ALONE (128-plane view 0.186 ms vs 0.199,
# 256: 0.380 vs 0.424, 512: 0.758 vs 0.804 -- profiles/r01_exp_r1zj_one_block_per_tile.log, r01_exp_r1zk_...), but its
# workgroups hold every CU until the launch ends, so the RCCL send/recv kernel cannot start beside it: in the rehearsal
# a rank of the 8-GPU run drops from 1200 to 1130 GStencil/s.  Short blocks retire every ~20 us and let it in.
def slab_options(workload, world, weak=False):
    opts = list(TUNED[workload])
    if workload == "c4" and world >= 4 and not weak:
        opts[opts.index("--sn") + 1] = "16"
    return opts


MIN_WARM_S = 0.25     # untimed warm-up continues (beyond --warmup steps) until the GPU has been busy this long
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md: 8.0 TB/s spec)


def kernel_arg_sets():
    """Kernels bench.py needs; prebuilt by __graft_entry__.build()."""
    return [TUNED[w] + [WORKLOADS[w]["stc"]] for w in ("c4", "c3", "c2", "c5")] + [STEP1[w] + [WORKLOADS[w]["stc"]] for w in ("c4", "c3")] + [TEMPORAL2[w] + [WORKLOADS[w]["stc"]] for w in ("c4", "c3")]


def pmc_traffic(workload, option_string):
    """HBM bytes per launch of this exact workload + kernel configuration from the committed rocprofv3 PMC
    passes (FETCH_SIZE x 2 + WRITE_SIZE, profiles/traffic_by_options.json); None when it was not profiled."""
    try:
        m = json.load(open(os.path.join(ROOT, "profiles", "traffic_by_options.json")))
        return m[workload][option_string]["traffic_bytes_per_launch"]
    except Exception:
        return None


def cpu_baseline(workload, step, budget_s=15.0):
    """Oracle (port) timed on the host cores on a bounded z/y-slab sample of the workload."""
    import numpy as np
    import oracle
    w = WORKLOADS[workload]
    spec = oracle.Spec(w["stc"], w["ndim"], step)
    L, M, N = spec.dims
    # bounded sample: a slab of the outermost dim (same plane size, same stencil, same dtype)
    if w["ndim"] == 3:
        Ls = min(L, 128)
        spec.set_dims(Ls, M, N)
        sample = "%d x %d x %d z-slab of the %d^3 grid" % (Ls, M, N, L)
    else:
        Ms = min(M, 4096)
        spec.set_dims(1, Ms, N)
        sample = "%d x %d y-slab" % (Ms, N)
    dt = np.float32 if w["dtype"] == "fp32" else np.float64
    rng = np.random.default_rng(1)
    A = rng.random(spec.shape, dtype=dt)
    B = np.zeros_like(A)
    h = spec.halo
    interior = 1
    for d in spec.shape:
        interior *= d - 2 * h
    oracle.sweep(spec, A, B, 1)  # warm (page faults)
    t0 = time.perf_counter()
    sweeps = 0
    while True:
        oracle.sweep(spec, A, B, 1)
        oracle.sweep(spec, B, A, 1)
        sweeps += 2
        el = time.perf_counter() - t0
        if el > budget_s or sweeps >= 400:
            break
    gst = sweeps * step * interior / el / 1e9
    return dict(value=gst, unit="GStencil/s", cores=oracle.threads(), kind="port",
                sample="%s, %d sweeps in %.1f s (OpenMP, %d threads)" % (sample, sweeps, el, oracle.threads()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS))
    ap.add_argument("--kernel-args", default=None, help="override the generator options (space separated)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = the workload's grid cut into N slabs (default, SURVEY 8e); weak = every rank holds a full-size slab (grid L*N planes)")
    ap.add_argument("--exchange-every", type=int, default=0, choices=[0, 1, 2],
                    help="N > 1: launches per halo exchange (2: ghost planes twice as wide, one exchange per ping-pong pair; "
                         "0 = multigpu.choose_exchange_every)")
    ap.add_argument("--headline-only", action="store_true", help="skip the side measurements (profiling runs: one dr_ kernel in the trace)")
    args = ap.parse_args()

    import torch
    import drstencil_amd as drs

    if not os.path.exists(drs.LIB_PATH):
        # a checkout without built artefacts: build the host library (g++ only, seconds); ranks take turns on a lock
        import fcntl
        import subprocess
        with open(os.path.join(ROOT, ".build.lock"), "w") as lk:
            fcntl.flock(lk, fcntl.LOCK_EX)
            if not os.path.exists(drs.LIB_PATH):
                subprocess.check_call(["make", "-C", os.path.join(ROOT, "drstencil_amd", "csrc")], stdout=subprocess.DEVNULL)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # DRS_REHEARSE="r/R": run the N > 1 code path as middle rank r of R on ONE GPU (process group of size 1, the rank is
    # its own neighbour) -- a rehearsal of the multi-GPU branch where only one GPU exists; the line says so
    rehearse = os.environ.get("DRS_REHEARSE")
    prank, pworld = (int(x) for x in rehearse.split("/")) if rehearse else (rank, world)
    assert not rehearse or world == 1
    # Generate + compile (or find cached) every kernel BEFORE HIP is initialised: a process
    # that has touched the GPU must not fork/exec the compiler.
    w = WORKLOADS[args.workload]
    opts = args.kernel_args.split() if args.kernel_args else (slab_options(args.workload, pworld, args.scaling == "weak") if pworld > 1 else TUNED[args.workload])
    spec = drs.Spec(w["stc"], w["ndim"], int(opts[opts.index("--step") + 1]) if "--step" in opts else 1)
    L, M, N = spec.dims
    H, step, iters = spec.halo, spec.step, spec.iterations
    weak = args.scaling == "weak" and pworld > 1
    if weak:       # the outermost dim grows with the number of ranks: fixed work per GPU
        if w["ndim"] == 3:
            L *= pworld
        else:
            M *= pworld
    kern1 = kernf = None
    if pworld == 1:
        kern = drs.Kernel(opts + [w["stc"]])
        if args.workload in STEP1 and not args.kernel_args and not args.headline_only:
            kern1 = drs.Kernel(STEP1[args.workload] + [w["stc"]])
            kernf = drs.Kernel(TEMPORAL2[args.workload] + [w["stc"]])
    else:
        from drstencil_amd.multigpu import HipSweep, SelfNeighbourRun, SlabPlan, SlabRun, measure_exchange_every
        auto_every = args.exchange_every == 0      # decided after the process group is up, from measured sweep / exchange times
        sweep = HipSweep(w["stc"], opts, os.path.join(ROOT, "drstencil_amd", "_kcache"))
        for ev in ((1, 2) if auto_every else (args.exchange_every,)):      # both modes' kernels: built (cache hits) before HIP is up
            sweep.prebuild(SlabPlan(L if w["ndim"] == 3 else M, H, pworld, prank, ev))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if pworld > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        from drstencil_amd.multigpu import nccl_options
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, pg_options=nccl_options(dist))
    assert world == args.gpus or world == 1 and args.gpus == 1, "launch with torch.distributed.run for --gpus > 1"

    tdt = torch.float32 if w["dtype"] == "fp32" else torch.float64
    esz = 4 if w["dtype"] == "fp32" else 8
    launches_per_step = spec.launches
    interior = (M - 2 * H) * (N - 2 * H) * ((L - 2 * H) if w["ndim"] == 3 else 1)
    npoints = M * N * (L if w["ndim"] == 3 else 1)

    ev_ms = 0.0
    warm_extra = 0
    calibration = None
    if pworld == 1:
        g = torch.Generator(device=dev).manual_seed(1)
        shape = (L, M, N) if w["ndim"] == 3 else (M, N)
        A = torch.rand(shape, dtype=tdt, device=dev, generator=g)
        B = torch.zeros_like(A)
        stream = torch.cuda.current_stream(dev)
        for _ in range(args.warmup):
            kern.run(A.data_ptr(), B.data_ptr(), stream=stream.cuda_stream)
        torch.cuda.synchronize()
        # a few milliseconds of launches do not bring the GPU to its steady clocks (C3: 221 us per launch right after 2 ms of
        # warm-up, 193 us from then on -- scripts/c3_loop_probe.py): keep running untimed steps until MIN_WARM_S have gone by
        tw = time.perf_counter()
        while args.warmup > 0 and time.perf_counter() - tw < MIN_WARM_S:
            for _ in range(8):
                kern.run(A.data_ptr(), B.data_ptr(), stream=stream.cuda_stream)
                warm_extra += 1
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(stream)
        n = 0
        for _ in range(args.steps):
            n += kern.run(A.data_ptr(), B.data_ptr(), stream=stream.cuda_stream)
        e1.record(stream)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        ev_ms = e0.elapsed_time(e1)
        assert n == launches_per_step * args.steps
        kinfo = kern.info
        kres = kern.resources
        parallelism = "1 GPU"
        def side(k, o, iters):
            # side measurement on the same grid (reference protocol: warm-up launches, then the timed
            # ping-pong loop bracketed by HIP events)
            n1, ms1 = k.run_timed(A.data_ptr(), B.data_ptr(), iterations=iters, warmup=4, stream=stream.cuda_stream)
            by = k.bytes_per_launch()
            return {"generator_options": " ".join(o), "GStencil_per_s": k.updates_per_launch() * n1 / (ms1 * 1e-3) / 1e9,
                    "avg_launch_ms": ms1 / n1, "achieved_GBps": by * n1 / (ms1 * 1e-3) / 1e9,
                    "roofline_frac": by * n1 / (ms1 * 1e-3) / 1e9 / HBM_PEAK_GBS}
        step1 = side(kern1, STEP1[args.workload], 16) if kern1 is not None else None
        fused2 = side(kernf, TEMPORAL2[args.workload], 32) if kernf is not None else None
    else:
        if auto_every:
            args.exchange_every, calibration = measure_exchange_every(torch, dist, (L, M, N) if w["ndim"] == 3 else (M, N), H, prank, pworld, sweep, dev, tdt,
                                                                       self_neighbour=bool(rehearse))
        run = (SelfNeighbourRun if rehearse else SlabRun)(torch, dist, (L, M, N) if w["ndim"] == 3 else (M, N), H, step, iters, prank, pworld, sweep, dev, tdt, every=args.exchange_every)
        g = torch.Generator(device=dev).manual_seed(1 + prank)
        run.A.copy_(torch.rand(run.A.shape, dtype=tdt, device=dev, generator=g))
        torch.cuda.synchronize()     # the run's streams are not the one that filled A
        for _ in range(args.warmup):
            run.run()
        torch.cuda.synchronize()
        nwarm = max(0, int(MIN_WARM_S / 0.003) * (1 if weak else pworld) - args.warmup) if args.warmup > 0 else 0   # ~MIN_WARM_S of runs; the same count on every rank (exchanges pair up)
        for _ in range(nwarm):
            run.run()
            warm_extra += 1
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(run.main)
        n = 0
        for _ in range(args.steps):
            n += run.run()
        e1.record(run.main)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        ev_ms = e0.elapsed_time(e1)
        t = torch.tensor([el, ev_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el, ev_ms = float(t[0]), float(t[1])
        kinfo = sweep.kernel(run.plan.interior[1] - run.plan.interior[0]).info
        kres = sweep.kernel(run.plan.interior[1] - run.plan.interior[0]).resources
        parallelism = "%s-slab x%d%s, RCCL send/recv halo every %d launch(es), overlapped" % ("z" if w["ndim"] == 3 else "y", pworld, " (weak: grid %s)" % "x".join(str(d) for d in ((L, M, N) if w["ndim"] == 3 else (M, N))) if weak else "", args.exchange_every)
        step1 = fused2 = None

    if rank == 0:
        total_launches = launches_per_step * args.steps
        updates = total_launches * step * interior
        if rehearse:    # one rank's share of the work only: the other ranks do not exist
            updates /= pworld
            parallelism = "REHEARSAL on one GPU of rank %d of %d (self-neighbour exchange through RCCL): %s" % (prank, pworld, parallelism)
        value = updates / el / 1e9
        # roofline of the dominant kernel dr_<name>: algorithmic bytes per launch / average
        # launch duration from the HIP events around the timed launches
        alg_bytes = 2.0 * esz * npoints / max(pworld, 1)    # per launch per GPU
        avg_launch_s = (ev_ms * 1e-3) / total_launches
        achieved = alg_bytes / avg_launch_s / 1e9
        out = {
            "metric": "GStencil/s (grid-point updates/s), 3d7pt_star" if args.workload in ("c3", "c4") else "GStencil/s (grid-point updates/s)",
            "value": value, "unit": "GStencil/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "warmup_steps_run": args.warmup + warm_extra,
            "ms_per_step": el * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak" if weak else "strong",
            "vs_baseline": None, "dtype": "f32" if w["dtype"] == "fp32" else "f64", "data": "synthetic",
            "config": {"workload": w["name"], "generator_options": " ".join(opts), "step": step,
                       "launches_per_step": launches_per_step, "parallelism": parallelism,
                       "kernel": "dr_" + kinfo["name"], "threads": kinfo["threads"], "lds_bytes": kinfo["lds_bytes"],
                       "vgprs": kres.get("vgprs"), "agprs": kres.get("agprs"), "scratch_bytes_per_lane": kres.get("scratch_bytes_per_lane"),
                       "occupancy_waves_per_simd": kres.get("occupancy_waves_per_simd")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(args.workload, " ".join(opts)) if pworld == 1 else None,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": avg_launch_s * 1e3},
        }
        if calibration:
            out["config"]["exchange_calibration"] = calibration      # measured on this machine during warm-up (multigpu.measure_exchange_every)
        out["step1_kernel"] = step1                 # one time step per launch: highest roofline fraction
        out["temporal_step2_kernel"] = fused2       # on-chip temporal blocking (2 stages): equal to the headline up to rounding
        if not args.no_cpu_baseline and pworld == 1:
            sys.path.insert(0, ROOT)
            out["cpu_baseline"] = cpu_baseline(args.workload, step)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
