#!/usr/bin/env python3
"""(out - in) mod 64 MiB for the kernels of an N > 1 run of C4: the exchanging launch's interior view and the whole-slab launch of ranks
of 8 and of 4 (bench.slab_options / slab_alone_options), forward and backward, delta = 0 .. 56 MiB in 8 MiB steps, warm."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import drstencil_amd as drs
from drstencil_amd.multigpu import SlabPlan, _write_view_stc

M = 1 << 20


def main():
    w = bench.WORKLOADS["c4"]
    cache = os.path.join(ROOT, "drstencil_amd", "_kcache")
    ks = []
    for world, rank in ((8, 3), (4, 1)):
        sp = SlabPlan(1024, 2, world, rank, 2)
        views = sorted(set(sp.views()))
        for lv in views:
            if lv < 32:
                continue
            ks.append(("w%d_view%d" % (world, lv), lv, drs.Kernel(bench.slab_options("c4", world) + [_write_view_stc(w["stc"], 3, lv, cache, "slabL")])))
        alone = bench.slab_alone_options("c4", world)      # None since the second half of round 3 (DRS_SLAB_ALONE_SN=1024 brings the one-block-per-tile kernel back)
        if alone:
            ks.append(("w%d_alone%d" % (world, sp.Lloc), sp.Lloc, drs.Kernel(alone + [_write_view_stc(w["stc"], 3, sp.Lloc, cache, "slabL")])))
    if os.environ.get("PROBE_BUILD_ONLY"):
        print([k[0] for k in ks])
        return
    import torch
    print("device uuid", getattr(torch.cuda.get_device_properties(0), "uuid", ""), flush=True)
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def one_way(k, a, b, n=12):
        k.launch(a, b, stream=st.cuda_stream)
        e0.record(st)
        for _ in range(n):
            k.launch(a, b, stream=st.cuda_stream)
        e1.record(st)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    nb = 272 * 4 * M            # room for the largest slab (264 planes + ghosts)
    arena = torch.empty((2 * nb + 80 * M) // 4, dtype=torch.float32, device="cuda")
    arena.uniform_()
    a, b0 = arena.data_ptr(), arena.data_ptr() + (nb + 64 * M - 1) // (64 * M) * (64 * M)
    tw = time.perf_counter()
    while time.perf_counter() - tw < 0.5:
        for _, _, k in ks:
            k.launch(a, b0, stream=st.cuda_stream); k.launch(b0, a, stream=st.cuda_stream)
        torch.cuda.synchronize()
    print("columns (us, fwd/bwd): " + ", ".join(n for n, _, _ in ks), flush=True)
    for d in range(0, 64, 8):
        line = "   out = in + k*64 MiB + %2d MiB:" % d
        for n, lv, k in ks:
            line += "   %.1f/%.1f" % (1e3 * one_way(k, a, b0 + d * M), 1e3 * one_way(k, b0 + d * M, a))
        print(line, flush=True)


main()
