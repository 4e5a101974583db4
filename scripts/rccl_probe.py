#!/usr/bin/env python3
"""rccl_probe.py -- what one rank of the 8-GPU C4 run costs per launch, measured on ONE GPU.

A single process initialises RCCL with world size 1 and plays rank 3 of 8 of the slab run: the real slab
kernels (two 6-plane boundary views + the 128-plane interior view), the real stream/event choreography of
drstencil_amd.multigpu.SlabRun and real RCCL batch_isend_irecv calls -- except that both neighbours are the
rank itself (periodic self-exchange: what goes "up" arrives in the lower ghost planes and vice versa), so
the bytes move through RCCL's self-copy path instead of an xGMI link.  It answers two questions the 1-GPU box
can answer: (1) does the overlap choreography work with RCCL's own streams, (2) is a rank host-bound (Python +
launch + RCCL enqueue cost per launch vs the GPU time per launch).  --graph replays the same loop from a HIP
graph (EXPERIMENT: torch's RCCL watchdog can abort on the captured events, profiles/r01_rccl_probe.md).

    python scripts/rccl_probe.py [--graph] [--steps 20]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--graph", action="store_true")
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--workload", default="c4")
    ap.add_argument("--every", type=int, default=2, help="launches per exchange (1 or 2)")
    a = ap.parse_args()
    import bench
    import drstencil_amd as drs
    from drstencil_amd.multigpu import HipSweep, SelfNeighbourRun, SlabPlan
    w = bench.WORKLOADS[a.workload]
    opts = bench.slab_options(a.workload, a.world)
    spec = drs.Spec(w["stc"], w["ndim"], 2)
    L, M, N = spec.dims
    H, step, iters = spec.halo, spec.step, spec.iterations
    sweep = HipSweep(w["stc"], opts, os.path.join(ROOT, "drstencil_amd", "_kcache"))
    frank = a.world // 2 - 1
    sweep.prebuild(SlabPlan(L, H, a.world, frank, a.every))

    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    from drstencil_amd.multigpu import nccl_options
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, pg_options=nccl_options(dist))

    run = SelfNeighbourRun(torch, dist, (L, M, N), H, step, iters, frank, a.world, sweep, dev, torch.float32, every=a.every)
    g = torch.Generator(device=dev).manual_seed(1)
    run.A.copy_(torch.rand(run.A.shape, dtype=torch.float32, device=dev, generator=g))
    A0 = run.A.clone()

    # correctness of the choreography: one launch A -> B, then B's ghost planes must hold the periodic images
    run.launch(run.A, run.B)
    torch.cuda.synchronize()
    p = run.plan
    ok_dn = torch.equal(run.B[p.recv_dn[0]:p.recv_dn[1]], run.B[p.send_up[0]:p.send_up[1]])
    ok_up = torch.equal(run.B[p.recv_up[0]:p.recv_up[1]], run.B[p.send_dn[0]:p.send_dn[1]])
    nz = bool((run.B[p.send_up[0]:p.send_up[1]] != 0).any())
    print("self-exchange after one launch: ghost_dn==sent_up %s, ghost_up==sent_dn %s, non-zero %s" % (ok_dn, ok_up, nz), flush=True)

    def timed(fn, steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        for _ in range(steps):
            n += fn()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        return n, t_host, time.perf_counter() - t0

    for _ in range(3):
        run.run()
    n, th, tt = timed(run.run, a.steps)
    print("eager : %d launches, host enqueue %.1f us/launch, wall %.1f us/launch" % (n, th / n * 1e6, tt / n * 1e6), flush=True)

    # the same rank without any exchange (pure kernels, same three views): the GPU time per launch
    run_world, run.world = run.world, 1
    n, th, tt = timed(run.run, a.steps)
    print("no-xchg: %d launches, host enqueue %.1f us/launch, wall %.1f us/launch" % (n, th / n * 1e6, tt / n * 1e6), flush=True)
    run.world = run_world

    if a.graph:
        run.A.copy_(A0)
        run.B.zero_()
        run.run()
        torch.cuda.synchronize()
        ref = run.A.clone()
        run.A.copy_(A0)
        run.B.zero_()
        cs = torch.cuda.Stream(device=dev)
        gr = torch.cuda.CUDAGraph()
        old_main = run.main
        run.main = cs
        with torch.cuda.graph(gr, stream=cs):
            run.run()
        run.main = old_main
        torch.cuda.synchronize()
        run.A.copy_(A0)
        run.B.zero_()
        gr.replay()
        torch.cuda.synchronize()
        print("graph replay equals eager: %s" % torch.equal(run.A, ref), flush=True)

        def rep():
            gr.replay()
            return spec.launches
        n, th, tt = timed(rep, a.steps)
        print("graph : %d launches, host enqueue %.1f us/launch, wall %.1f us/launch" % (n, th / n * 1e6, tt / n * 1e6), flush=True)
        del gr      # a live graph holding RCCL nodes makes destroy_process_group wait forever
        torch.cuda.synchronize()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
