"""Shared helpers for the tests: golden-case loading, .stc writing."""
import glob
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def golden_cases():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz")))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    return meta, z["a0"], z["a"], z["b"]


def write_stc(path, ndim, dims, iters, pts, iter_token="iterations"):
    L, M, N = dims
    with open(path, "w") as f:
        if ndim == 3:
            f.write("L %d\n" % L)
        f.write("M %d\nN %d\n\n%s %d\n\nstencil\n" % (M, N, iter_token, iters))
        for p in pts:
            f.write(" ".join(repr(v) if isinstance(v, float) else str(v) for v in p) + "\n")


def stc_from_meta(tmpdir, meta, name="g"):
    path = os.path.join(str(tmpdir), name + ".stc")
    write_stc(path, meta["ndim"], tuple(meta["dims"]), meta["iterations"], [tuple(p) for p in meta["points"]])
    return path
