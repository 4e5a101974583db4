#!/usr/bin/env python3
"""prebuild_space.py <stc> <fp32|fp64> <order> <steps> [--3d] -- compile every configuration of the tuner's space for one
stencil into drstencil_amd/_kcache on THIS machine (no GPU needed), so that a later tuner run on the GPU box only measures."""
import os
import sys
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from drstencil_amd.tuner import tuning as t


def main():
    stc, dtype, order, steps = sys.argv[1], sys.argv[2], int(sys.argv[3]), tuple(int(s) for s in sys.argv[4].split(","))
    is3d = "--3d" in sys.argv
    t.order, t.ndim, t.elem_bytes = order, 3 if is3d else 2, 4 if dtype == "fp32" else 8
    jobs = []
    for v in t.enumerate_space(steps):
        jobs.append((t.cfgToString(v), (["--3d"] if is3d else []) + ["--dtype", dtype] + t.cfgToCommandLine(v).split() + [os.path.abspath(stc)]))
    with ProcessPoolExecutor(max_workers=int(os.environ.get("JOBS", "8"))) as ex:
        res = list(ex.map(t._build, jobs, chunksize=4))
    print("%s: %d configurations, %d built, %d rejected" % (os.path.basename(stc), len(jobs), sum(1 for r in res if r[1]), sum(1 for r in res if not r[1])))


if __name__ == "__main__":
    main()
