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


def rows_from_tuner_log(t, log, stc, is3d, dtype, steps, seed, max_configs, order=1):
    """Labelled rows from a tuner run on the GPU box (profiles/r02_tune_*.txt): the run is re-enumerated with its seed, every
    configuration it compiled is labelled by its outcome in the log (timed = clean, 'exceeds the register file' = spill) and
    its features come from the generator (no compilation)."""
    import random as _r
    import re
    t.order, t.ndim, t.elem_bytes = order, 3 if is3d else 2, 4 if dtype == "fp32" else 8
    paras = t.enumerate_space(steps)
    _r.seed(seed)
    _r.shuffle(paras)
    paras = paras[:max_configs]
    outcome = {}
    for line in open(log):
        m = re.match(r"\d+/\d+: (\S+)\s+(BUILD FAILED.*register file|\d+ ns)", line)
        if m:
            outcome[m.group(1)] = 1 if m.group(2).startswith("BUILD") else 0
    rows = []
    for v in paras:
        name = t.cfgToString(v)
        if name not in outcome:
            continue
        args = (["--3d"] if is3d else []) + ["--dtype", dtype] + t.cfgToCommandLine(v).split() + [os.path.join(ROOT, stc)]
        info = t.kernel_info(args)
        if info is None:
            continue
        rows.append(dict(args=" ".join(args[:-1]), threads=info["threads"], demand=info["reg_demand"], schedule=info["schedule"], taps=info["taps"],
                         pts=info["points_per_lane"], stages=info["stages"], words=1 if dtype == "fp32" else 2, vgprs=0, agprs=0, scratch=outcome[name], source=os.path.basename(log)))
    return rows


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
    cached = os.path.join(ROOT, "profiles", "r02_reg_model_rows.jsonl")
    if os.environ.get("REGFIT_REUSE") == "1" and os.path.exists(cached):      # re-fit without recompiling the sample
        rows = [r for r in (json.loads(l) for l in open(cached)) if "source" not in r]
    else:
        with ProcessPoolExecutor(max_workers=os.cpu_count()) as ex:
            rows = list(ex.map(build, jobs, chunksize=2))
    # outcomes of tuner runs on the GPU box widen the sample where the tuner actually searches
    for log, stc, is3d, dtype, steps, seed, mx in (("r02_tune_c4_s2_reuse_space.txt", "benchmarks/configs/c4_3d7pt_star_1024.stc", True, "fp32", (2,), 7, 1100),):
        lp = os.path.join(ROOT, "profiles", log)
        if os.path.exists(lp):
            rows += rows_from_tuner_log(t, lp, stc, is3d, dtype, steps, seed, mx)
    ok = [r for r in rows if "error" not in r and r["vgprs"] >= 0]
    clean = [r for r in ok if r["scratch"] == 0]
    spill = [r for r in ok if r["scratch"] > 0]
    # logistic model of "spills to scratch" on what the generator knows before compiling (tuning.spill_features)
    import numpy as np
    F = np.array([t.spill_features(r["demand"], r["threads"], r["taps"], r["pts"], r["stages"], r["words"], r["schedule"], t.prefetch_depth_of(r["args"].split())) + [1.0] for r in ok])
    Y = np.array([1.0 if r["scratch"] > 0 else 0.0 for r in ok])
    mu, sd = F.mean(0), F.std(0)
    mu[-1], sd[-1] = 0.0, 1.0
    sd[sd == 0] = 1.0
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
        f.write("Features (all known before compiling, from the generator's kernel info and the option vector; tuning.spill_features): b = registers a lane can "
                "have at the workgroup's size (512 / waves per SIMD); x1 = reg_demand / b; x2 = words * taps * points_per_lane * stages / b (FMAs per plane); "
                "x3 = temporal; x4 = fp64; x5 = words * points_per_lane / b; x6 = workgroup of <= 256 lanes (the AGPR half is usable); x7 = reuse schedule; "
                "x8 = prefetch depth / 3; x9 = x1^2; x10 = x1*x2; x11 = x1*x6; x12 = x1*x3.\n\n")
        # holdout check: fit on 70 %, evaluate on the other 30 %
        rng = np.random.default_rng(0)
        idx = rng.permutation(len(ok)); ntr = int(0.7 * len(ok))
        w2 = np.zeros(F.shape[1])
        mu2, sd2 = F[idx[:ntr]].mean(0), F[idx[:ntr]].std(0); mu2[-1], sd2[-1] = 0.0, 1.0; sd2[sd2 == 0] = 1.0
        Fn2 = (F[idx[:ntr]] - mu2) / sd2
        for _ in range(30000):
            p2 = 1.0 / (1.0 + np.exp(-Fn2 @ w2)); w2 -= 0.5 * (Fn2.T @ (p2 - Y[idx[:ntr]])) / ntr
        pt = 1.0 / (1.0 + np.exp(-(((F[idx[ntr:]] - mu2) / sd2) @ w2))); yt = Y[idx[ntr:]]
        pred = pt > 0.35
        fn = int((~pred & (yt == 1)).sum()); tn = int((~pred & (yt == 0)).sum()); fp = int((pred & (yt == 0)).sum())
        f.write("Holdout (fit on 70 %%, threshold 0.35 on the other %d): %.1f %% of the kept configurations spill, %.1f %% of the usable ones are lost.\n\n"
                % (len(yt), 100.0 * fn / max(fn + tn, 1), 100.0 * fp / max(fp + tn, 1)))
        f.write("Logistic fit: P(spill) = sigmoid(%.4f + %s).\n\n" % (bias, " + ".join("%.4f*x%d" % (wt, i + 1) for i, wt in enumerate(weights))))
        f.write("| drop when P(spill) > | spills dropped | spills kept (compiled, then refused) | clean dropped | clean kept | spill rate among kept | usable configurations lost |\n|---|---|---|---|---|---|---|\n")
        for thr in (0.25, 0.35, 0.5, 0.6, 0.7, 0.8):
            pred = p > thr
            tp = int((pred & (Y == 1)).sum()); fp = int((pred & (Y == 0)).sum()); fn = int((~pred & (Y == 1)).sum()); tn = int((~pred & (Y == 0)).sum())
            f.write("| %.2f | %d | %d | %d | %d | %.1f %% | %.1f %% |\n" % (thr, tp, fn, fp, tn, 100.0 * fn / max(fn + tn, 1), 100.0 * fp / max(fp + tn, 1)))
        f.write("\nWithout the filter %.1f %% of the compiled configurations are refused for scratch (round 1: 45 %% of the random 3D ones). "
                "tuning.py uses the threshold 0.35.\n" % (100.0 * len(spill) / max(len(ok), 1)))
    print(open(out).read())
    print(json.dumps(dict(SPILL_BIAS=round(bias, 4), SPILL_W=[round(x, 4) for x in weights])))
    with open(os.path.join(ROOT, "profiles", "r02_reg_model_rows.jsonl"), "w") as f:
        for r in rows:
            f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
