#!/usr/bin/env python3
"""Launch time vs (output array - input array) over a wide range, warm buffers (0.4 s of launches first): one arena, output at input +
array bytes + delta, delta = 0 .. 200 MiB in 4 MiB steps (C4: step-2 headline, step 1 full rows, fused step 3; C3: headline, fused step 3).
The two directions of the ping-pong are timed SEPARATELY (in -> out at +delta, and out -> in at -delta)."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import drstencil_amd as drs

M, G = 1 << 20, 1 << 30


def main():
    sets = {}
    for wl in ("c4", "c3"):
        w = bench.WORKLOADS[wl]
        sets[wl] = [("s2", drs.Kernel(list(bench.TUNED[wl]) + [w["stc"]])), ("s1", drs.Kernel(list(bench.STEP1[wl]) + [w["stc"]])), ("s3", drs.Kernel(list(bench.FUSED3[wl][0]) + [w["stc"]]))]
    if os.environ.get("PROBE_BUILD_ONLY"):
        return
    import torch
    print("device uuid", getattr(torch.cuda.get_device_properties(0), "uuid", ""), flush=True)
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def one_way(k, a, b, n=6):
        k.launch(a, b, stream=st.cuda_stream)
        e0.record(st)
        for _ in range(n):
            k.launch(a, b, stream=st.cuda_stream)
        e1.record(st)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    for wl, ks in sets.items():
        i = ks[0][1].info
        nb = i["L"] * i["M"] * i["N"] * 4
        arena = torch.empty((2 * nb + 300 * M) // 4, dtype=torch.float32, device="cuda")
        arena.uniform_()
        base = (arena.data_ptr() + 8 * M) // (2 * M) * (2 * M)
        tw = time.perf_counter()
        while time.perf_counter() - tw < 0.5:
            for _, k in ks:
                k.launch(base, base + nb + 32 * M, stream=st.cuda_stream); k.launch(base + nb + 32 * M, base, stream=st.cuda_stream)
            torch.cuda.synchronize()
        print("== %s (array %d MiB, plane %.1f MiB): forward = in -> out at in + array + delta; backward = the reverse launch of the ping-pong" % (wl, nb >> 20, nb / i["L"] / M), flush=True)
        for d in list(range(0, 204, 4)):
            line = "   delta %3d MiB:" % d
            for n, k in ks:
                line += "   %s fwd %.4f bwd %.4f" % (n, one_way(k, base, base + nb + d * M), one_way(k, base + nb + d * M, base))
            print(line, flush=True)
        del arena
        torch.cuda.empty_cache()


main()
