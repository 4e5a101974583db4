#!/usr/bin/env python3
"""Randomised sweep through the CPU emulator (no GPU, no GPU compiler): fuzz_emulated.py <n> <seed> [jobs].
The larger sibling of tests/test_emulated_kernels.py::test_emulated_sampled_fuzz."""
import json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from concurrent.futures import ProcessPoolExecutor


def one(job):
    import oracle
    from emu_util import build_emulated, run_emulated
    from helpers import write_stc
    import test_emulated_kernels as tk
    vid, ndim, pts, dims, opts, step = job
    tmp = tempfile.mkdtemp()
    stc = os.path.join(tmp, "f.stc")
    write_stc(stc, ndim, dims, 4, getattr(tk._mg(), pts))
    try:
        lib = build_emulated(tmp, stc, opts)
    except AssertionError:
        return vid, "rejected"
    spec = oracle.Spec(stc, ndim, step)
    dt = np.float32 if "fp32" in opts else np.float64
    A = oracle.fill_random(spec.shape, dt); B = np.zeros_like(A)
    A2, B2 = A.copy(), B.copy()
    oracle.run(spec, A2, B2, contract=1)
    run_emulated(lib, A, B, spec.iterations, step)
    if json.loads(lib.drs_plugin_info().decode()).get("stages", 1) > 1:
        bar = 1e-6 if dt == np.float32 else 1e-12
        ok = oracle.check(spec, A, A2)["max_rel"] < bar and oracle.check(spec, B, B2)["max_rel"] < bar
    else:
        ok = np.array_equal(A, A2) and np.array_equal(B, B2)
    return vid, "ok" if ok else "MISMATCH " + " ".join(opts)


def main():
    import test_emulated_kernels as tk
    n, seed = int(sys.argv[1]), int(sys.argv[2])
    jobs = tk._emulated_fuzz_jobs(n, seed)
    with ProcessPoolExecutor(int(sys.argv[3]) if len(sys.argv) > 3 else 7) as ex:
        res = list(ex.map(one, jobs))
    bad = [r for r in res if r[1].startswith("MISMATCH")]
    for r in bad:
        print(r[0], r[1])
    print("%d configurations: %d ok, %d rejected by the generator, %d MISMATCHES" % (len(res), sum(r[1] == "ok" for r in res), sum(r[1] == "rejected" for r in res), len(bad)))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
