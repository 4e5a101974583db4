#!/usr/bin/env python3
"""fit_reg_model.py [n_per_space] -- fit the tuner's register model (drstencil_amd/tuner/tuning.py: REG_A, REG_B, REG_SLACK).

Random configurations of the 3D (fp32, fp64) and 2D tuning spaces are generated and cross-compiled for gfx950 (no GPU
needed: hipcc's -Rpass-analysis=kernel-resource-usage report is the ground truth).  For each one: the generator's
`reg_demand` (32-bit registers of the per-lane state it names), the workgroup size, and the compiler's VGPRs + AGPRs and
scratch bytes.  A logistic model of "spills to scratch" is fitted on features the generator knows before compiling
(tuning.spill_features) and tabulated for several thresholds.  Writes profiles/r02_reg_model.md (+ the raw rows)."""
import json
import os
import random
import sys
import tempfile
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["DRS_ALLOW_SCRATCH"] = "1"


def build(job):
    import drstencil_amd as drs
    args, cache = job
    try:
        k = drs.Kernel(args, cache_dir=cache)
        return dict(args=" ".join(args[:-1]), threads=k.info["threads"], demand=k.info["reg_demand"], schedule=k.info["schedule"],
                    taps=k.info["taps"], pts=k.info["points_per_lane"], stages=k.info["stages"], words=1 if k.info["dtype"] == "fp32" else 2,
                    vgprs=k.resources["vgprs"], agprs=k.resources["agprs"], scratch=k.resources["scratch_bytes_per_lane"])
    except Exception as e:
        return dict(args=" ".join(args[:-1]), error=str(e)[-200:])


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    from drstencil_amd.tuner import tuning as t
    cache = tempfile.mkdtemp(prefix="regfit_")
    jobs = []
    random.seed(11)
    for ndim, dtype, stc, steps in ((3, "fp32", "benchmarks/configs/c3_3d7pt_star_512.stc", (1, 2, 3)), (3, "fp64", "benchmarks/3d7pt_star/3d7pt_star.stc", (1, 2)),
                                    (3, "fp64", "benchmarks/3d9pt_cross/3d9pt_cross.stc", (2,)), (2, "fp64", "benchmarks/2d25pt_box/2d25pt_box.stc", (1, 2)),
                                    (2, "fp32", "benchmarks/configs/c2_2d5pt_star_8192.stc", (1, 2))):
        t.ndim, t.elem_bytes = ndim, 4 if dtype == "fp32" else 8
        t.order = 2 if ("cross" in stc or "25pt" in stc) else 1
        space = t.enumerate_space(steps)
        for v in random.sample(space, min(n, len(space))):
            jobs.append(((["--3d"] if ndim == 3 else []) + ["--dtype", dtype] + t.cfgToCommandLine(v).split() + [os.path.join(ROOT, stc)], cache))
    with ProcessPoolExecutor(max_workers=os.cpu_count()) as ex:
        rows = list(ex.map(build, jobs, chunksize=2))
    ok = [r for r in rows if "error" not in r and r["vgprs"] >= 0]
    clean = [r for r in ok if r["scratch"] == 0]
    spill = [r for r in ok if r["scratch"] > 0]
    # logistic model of "spills to scratch" on what the generator knows before compiling (tuning.spill_features)
    import numpy as np
    F = np.array([t.spill_features(r["demand"], r["threads"], r["taps"], r["pts"], r["stages"], r["words"]) + [1.0] for r in ok])
    Y = np.array([1.0 if r["scratch"] > 0 else 0.0 for r in ok])
    mu, sd = F.mean(0), F.std(0)
    mu[-1], sd[-1] = 0.0, 1.0
    Fn = (F - mu) / sd
    w = np.zeros(F.shape[1])
    for _ in range(30000):
        p = 1.0 / (1.0 + np.exp(-Fn @ w))
        w -= 0.5 * (Fn.T @ (p - Y)) / len(Y)
    wr = w / sd
    bias = float(w[-1] - (w[:-1] * mu[:-1] / sd[:-1]).sum())
    weights = [float(x) for x in wr[:-1]]
    p = 1.0 / (1.0 + np.exp(-(F[:, :-1] @ np.array(weights) + bias)))
    out = os.path.join(ROOT, "profiles", "r02_reg_model.md")
    with open(out, "w") as f:
        f.write("# Register model of the tuner's registerFilter (scripts/fit_reg_model.py)\n\n")
        f.write("%d random configurations of the 3D (fp32 / fp64, steps 1-3, scatter and reuse schedules, temporal) and 2D spaces, cross-compiled for gfx950; "
                "ground truth = hipcc's kernel-resource-usage report (ScratchSize > 0 = refused by the runtime).\n\n" % len(rows))
        f.write("* generator / runtime errors (invalid geometry etc.): %d\n* build without scratch: %d\n* spill to scratch: %d (%.0f %% of the compiled ones)\n\n"
                % (len(rows) - len(ok), len(clean), len(spill), 100.0 * len(spill) / max(len(ok), 1)))
        f.write("Features (all known before compiling, from the generator's kernel info): b = registers a lane can have at the workgroup's size "
                "(512 / waves per SIMD); x1 = reg_demand / b; x2 = words * taps * points_per_lane * stages / b (FMAs per plane); x3 = temporal; "
                "x4 = fp64; x5 = words * points_per_lane / b.\n\n")
        f.write("Logistic fit: P(spill) = sigmoid(%.4f + %s).\n\n" % (bias, " + ".join("%.4f*x%d" % (wt, i + 1) for i, wt in enumerate(weights))))
        f.write("| drop when P(spill) > | spills dropped | spills kept (compiled, then refused) | clean dropped | clean kept | spill rate among kept | usable configurations lost |\n|---|---|---|---|---|---|---|\n")
        for thr in (0.3, 0.4, 0.5, 0.6, 0.7, 0.8):
            pred = p > thr
            tp = int((pred & (Y == 1)).sum()); fp = int((pred & (Y == 0)).sum()); fn = int((~pred & (Y == 1)).sum()); tn = int((~pred & (Y == 0)).sum())
            f.write("| %.1f | %d | %d | %d | %d | %.1f %% | %.1f %% |\n" % (thr, tp, fn, fp, tn, 100.0 * fn / max(fn + tn, 1), 100.0 * fp / max(fp + tn, 1)))
        f.write("\nWithout the filter %.1f %% of the compiled configurations are refused for scratch (round 1: 45 %% of the random 3D ones). "
                "tuning.py uses the threshold 0.6.\n" % (100.0 * len(spill) / max(len(ok), 1)))
    print(open(out).read())
    print(json.dumps(dict(SPILL_BIAS=round(bias, 4), SPILL_W=[round(x, 4) for x in weights])))
    with open(os.path.join(ROOT, "profiles", "r02_reg_model_rows.jsonl"), "w") as f:
        for r in rows:
            f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
