#!/usr/bin/env python3
"""Differential sweep against the REFERENCE ITSELF over random stencil shapes (authoring container only: it runs
oracle/_ref/drstencil_ref, the reference generator built from /root/reference by oracle/Makefile; nothing of it is stored).

TEST INFRASTRUCTURE, the wide-angle companion of oracle/make_golden.py.  For every random shape (tests/fuzz_shapes.py:
sparse to dense, one-sided, without a centre, duplicate offsets, mixed signs, orders 1-3) x --step 1..3 x --dist
(automatic, inside and outside the legal range):

  front end   the reference binary and bin/drstencil on the same command line: exit code, stdout, and -- when both emit --
              the macro block (L M N Iterations Range Halo Dist) and the gold kernel's term list (offsets in order, the
              coefficient literals as printed);  drstencil_amd.Spec (the C-ABI view of the IR) must say the same;
  numerics    for a sample of the cases the reference-emitted gold statement is lifted into a throw-away C++ driver (as
              make_golden.py does, g++ -O0, the reference's own rand() fill) and its output arrays must equal the CPU
              oracle's (oracle.run, contract=0) bit for bit: the oracle is pinned on shapes nobody drew by hand.

usage: fuzz_vs_reference.py <shapes> <seed> [numeric cases].  Prints one summary line; exit code 1 on any difference."""
import os, random, re, subprocess, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, HERE)
import numpy as np
import make_golden as mg
import oracle
import drstencil_amd as drs
import fuzz_shapes as fs

CLI = os.path.join(REPO, "bin", "drstencil")
MACROS = ("L", "M", "N", "Iterations", "Range", "Halo", "Dist")


def ours(cwd, args):
    # stdout is the contract (the reference writes nothing else); our one-line reason for a rejection goes to stderr
    p = subprocess.run([CLI, "--ref-defaults"] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
    ours.reason = p.stderr.strip()
    return (p.returncode if p.returncode >= 0 else 256 + p.returncode), p.stdout


def numeric(tmp, ndim, macros, guard, stmt, inc, dims, stc, step):
    """The reference-emitted gold statement run on the reference's own input fill == the oracle, bit for bit."""
    defs = "#define NDIM %d\n" % ndim + "".join("#define %s %d\n" % kv for kv in macros.items() if kv[0] not in ("Bx", "By", "Sn"))
    if ndim == 2 and "L" not in macros:
        defs += "#define L 1\n"
    with open(os.path.join(tmp, "drv.cpp"), "w") as f:
        f.write(mg.DRIVER % dict(ref=mg.REF, defs=defs, guard=guard, stmt=stmt, inc=inc))
    subprocess.check_call(["g++", "-O0", "-std=c++17", "-o", "drv", "drv.cpp"], cwd=tmp)
    launches = int(subprocess.check_output(["./drv"], cwd=tmp, text=True).strip())
    shape = dims if ndim == 3 else dims[1:]
    ref = {k: np.fromfile(os.path.join(tmp, k + ".bin"), dtype=np.float64).reshape(shape) for k in ("a0", "a", "b")}
    spec = oracle.Spec(stc, ndim, step)
    A = oracle.fill_random(shape, np.float64)
    B = np.zeros_like(A)
    n = oracle.run(spec, A, B, contract=0)
    return n == launches and np.array_equal(ref["a0"], oracle.fill_random(shape, np.float64)) and np.array_equal(A, ref["a"]) and np.array_equal(B, ref["b"])


def main():
    nshapes = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    nnum = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    rnd = random.Random(seed)
    tmp = tempfile.mkdtemp(prefix="drs_vs_ref_")
    cases = both_emit = both_reject = numeric_ok = numeric_n = lds_limit = 0
    diffs = []
    for s in range(nshapes):
        ndim = rnd.choice([2, 3])
        h = rnd.choice([1, 1, 2] if ndim == 3 else [1, 2, 2, 3])
        pts, _mixed = fs.random_shape(rnd, ndim, h)
        # the reference's default tile is 16 x 16 (x 16): grids a few tiles wide keep the numeric driver cheap
        dims = (rnd.randint(6 * h + 2, 6 * h + 9), rnd.randint(6 * h + 2, 24), rnd.randint(6 * h + 2, 40)) if ndim == 3 else (1, rnd.randint(6 * h + 2, 50), rnd.randint(6 * h + 2, 70))
        stc = os.path.join(tmp, "g.stc")
        mg.write_stc(stc, ndim, dims, rnd.randint(1, 7), pts, iter_token="iterations" if rnd.random() < 0.9 else "iteratioins")
        for step in (1, 2, 3):
            if len(set(p[:-1] for p in pts)) ** step > 2000:
                continue
            for dist in {0, rnd.randint(1, 2 * h * step + 1), rnd.randint(max(1, (step - 1) * h), step * h)}:
                geo = []
                if rnd.random() < 0.5:      # the reference's own geometry options: they decide "Invalid configuration!" and the Bx / By / Sn macros
                    for opt, vals in (("--bx", (8, 16, 32, 64)), ("--by", (2, 4, 8, 16)), ("--sn", (4, 16, 64)), ("--stream-unroll", (1, 2, 4)),
                                      (rnd.choice(["--block-merge-x", "--cyclic-merge-x"]), (1, 2, 4)), (rnd.choice(["--block-merge-y", "--cyclic-merge-y"]), (1, 2, 4)),
                                      ("--merge-forward", (0, 2, 5, 100))):
                        if rnd.random() < 0.4:
                            geo += [opt, str(rnd.choice(vals))]
                    if rnd.random() < 0.3:
                        geo.append("--prefetch")
                    if ndim == 2 and rnd.random() < 0.4:
                        geo.append("--streaming")
                args = (["--3d"] if ndim == 3 else []) + ["--step", str(step)] + (["--dist", str(dist)] if dist else []) + geo + ["--check", "-o", "g.cu", "g.stc"]
                for f in ("g.cu",):
                    if os.path.exists(os.path.join(tmp, f)):
                        os.remove(os.path.join(tmp, f))
                rrc, rout = mg.run_ref(tmp, args)
                rsrc = open(os.path.join(tmp, "g.cu")).read() if os.path.exists(os.path.join(tmp, "g.cu")) else None
                if rsrc is not None:
                    os.remove(os.path.join(tmp, "g.cu"))
                orc, oout = ours(tmp, args)
                osrc = open(os.path.join(tmp, "g.cu")).read() if os.path.exists(os.path.join(tmp, "g.cu")) else None
                cases += 1
                tag = "%dd o%d shape %d: %s" % (ndim, h, s, " ".join(args[:-4]))
                if rrc == "hang":
                    continue
                if rrc == 0 and rsrc is not None and orc == 255 and oout == "Invalid configuration!\n" and "160 KiB of LDS" in ours.reason:
                    lds_limit += 1          # the one intended divergence: the reference never checks its tile against the shared-memory
                    continue                # capacity (its program would fail at launch); the planner checks MI355X's 160 KiB and says so
                if (rrc, rout, rsrc is not None) != (orc, oout, osrc is not None):
                    diffs.append("%s: CLI differs: reference rc=%s emitted=%s %r | ours rc=%s emitted=%s %r" % (tag, rrc, rsrc is not None, rout[-120:], orc, osrc is not None, oout[-120:] + (" [stderr: %s]" % ours.reason if ours.reason else "")))
                    continue
                if rsrc is None:
                    both_reject += 1
                    continue
                both_emit += 1
                macros, guard, stmt, inc, terms = mg.lift(rsrc, "g")
                omac = {m.group(1): int(m.group(2)) for m in re.finditer(r"^#define (L|M|N|Iterations|Range|Halo|Dist|Bx|By|Sn) (-?\d+)\s*$", osrc, re.M)}
                # 2D streaming ignores --by (codegen_2d.hpp:125); we report the effective 1
                bad = [k for k in MACROS + ("Bx", "By", "Sn") if k in macros and omac.get(k) != macros[k] and not (k == "By" and "--streaming" in args and ndim == 2)]
                if bad:
                    diffs.append("%s: macros differ: %s reference %s ours %s" % (tag, bad, {k: macros[k] for k in bad}, {k: omac.get(k) for k in bad}))
                    continue
                spec = drs.Spec(stc, ndim, step, dist)
                mine = [(list(off[3 - ndim:]), text) for off, _c, text in spec.points]
                theirs = [(t["off"], t["coef"]) for t in terms]
                if mine != theirs or spec.halo != macros["Halo"] or spec.dist != macros["Dist"] or ("Range" in macros and spec.range != macros["Range"]):
                    diffs.append("%s: gold terms / IR differ (%d vs %d terms)" % (tag, len(mine), len(theirs)))
                    continue
                if numeric_n < nnum and rnd.random() < 0.35:
                    numeric_n += 1
                    if numeric(tmp, ndim, macros, guard, stmt, inc, dims, stc, step):
                        numeric_ok += 1
                    else:
                        diffs.append("%s: oracle output differs from the reference-emitted gold statement" % tag)
    print("%d shapes, %d command lines against the reference binary: %d emitted by both (macros, gold term order and coefficient literals identical), "
          "%d rejected by both with the same message and exit code, %d rejected by us alone for MI355X's LDS capacity (the reference does not check its tile against shared memory); oracle == reference-emitted gold statement bit for bit on %d of %d sampled cases; %d DIFFERENCES"
          % (nshapes, cases, both_emit, both_reject, lds_limit, numeric_ok, numeric_n, len(diffs)))
    for d in diffs[:int(os.environ.get("DIFFS_SHOWN", "40"))]:
        print("  ", d)
    sys.exit(1 if diffs else 0)


if __name__ == "__main__":
    main()
