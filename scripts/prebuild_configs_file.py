#!/usr/bin/env python3
"""prebuild_configs_file.py <configs file> [<3D .stc>] -- compile every configuration of a tuner `--configs-file` (one raw option string per
line, as scripts/archive/tune_c4_s2_top.txt) for an fp32 3D spec into drstencil_amd/_kcache on THIS machine (no GPU needed), with the flags the
round-3 sweeps use (`--3d --dtype fp32 --cc-opt -fno-slp-vectorize` in front), so that the tuner run on the GPU box only measures."""
import os
import sys
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from drstencil_amd.tuner import tuning as t


def main():
    stc = os.path.abspath(sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "benchmarks", "configs", "c4_3d7pt_star_1024.stc"))
    jobs = []
    for line in open(sys.argv[1]):
        line = line.strip()
        if not line or line.startswith("#"):
            continue
        jobs.append((line.replace(" ", "").replace("--", "_"), ["--3d", "--dtype", "fp32", "--cc-opt", "-fno-slp-vectorize"] + line.split() + [stc]))
    with ProcessPoolExecutor(max_workers=int(os.environ.get("JOBS", "8"))) as ex:
        res = list(ex.map(t._build, jobs, chunksize=2))
    print("%d configurations, %d built, %d rejected or refused" % (len(jobs), sum(1 for r in res if r[1]), sum(1 for r in res if not r[1])))


if __name__ == "__main__":
    main()
