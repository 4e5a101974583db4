#!/usr/bin/env python3
"""Generate tests/golden/ fixtures from the REFERENCE ITSELF (run in the authoring
container only; /root/reference does not exist on the GPU box).

TEST INFRASTRUCTURE.  What it does, per case:
  1. runs oracle/_ref/drstencil_ref (the reference generator, built by oracle/Makefile
     from /root/reference/main.cpp where it lies) with --check on a small .stc written
     to a temp dir -> the reference-emitted .cu text;
  2. lifts out of that text, verbatim: the #define block (L M N Iterations Range Halo
     Dist), the gold_<name> interior guard + statement (codegen.hpp:654-656) and the
     ping-pong loop increment (codegen.hpp:609);
  3. wraps the lifted statement in a plain loop nest in a throw-away C++ file that
     #includes the reference's common.hpp from /root/reference for the input fill
     (common.hpp:9-45), compiles it with `g++ -O0` and runs it;
  4. stores inputs and outputs (data only) as tests/golden/<case>.npz, and the CLI
     behaviour (stdout, exit code, macros, gold terms) in tests/golden/ref_cli.json.

No reference source text is stored: fixtures hold numbers, macro values, the list of
(offset, coefficient-as-printed) gold terms and CLI messages.
"""
import json
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.environ.get("DRS_REFERENCE", "/root/reference")
REFBIN = os.path.join(HERE, "_ref", "drstencil_ref")
GOLD = os.path.join(REPO, "tests", "golden")

STAR3 = [(0, 0, 0, 0.3), (1, 0, 0, 0.2), (-1, 0, 0, 0.2), (0, 1, 0, 0.2), (0, -1, 0, 0.2), (0, 0, 1, 0.2), (0, 0, -1, 0.2)]
CROSS3 = [(0, 0, 0, 0.3)] + [(a, b, c, 0.2) for a in (1, -1) for b in (1, -1) for c in (1, -1)]
STAR2 = [(0, 0, 0.3), (0, 1, 0.2), (1, 0, 0.2), (0, -1, 0.2), (-1, 0, 0.2)]
CROSS2 = [(0, 0, 0.3), (1, -1, 0.2), (-1, 1, 0.2), (1, 1, 0.2), (-1, -1, 0.2)]
BOX9 = [(0, 0, 0.3), (1, 0, 0.2), (-1, 0, 0.2), (0, 1, 0.2), (0, -1, 0.2), (1, 1, 0.1), (-1, 1, 0.1), (1, -1, 0.1), (-1, -1, 0.1)]
STAR9 = [(0, 0, 0.3), (1, 0, 0.2), (-1, 0, 0.2), (0, 1, 0.2), (0, -1, 0.2), (2, 0, 0.1), (-2, 0, 0.1), (0, 2, 0.1), (0, -2, 0.1)]
CROSS9 = [(0, 0, 0.3), (1, 1, 0.2), (1, -1, 0.2), (-1, 1, 0.2), (-1, -1, 0.2), (2, 2, 0.1), (2, -2, 0.1), (-2, 2, 0.1), (-2, -2, 0.1)]
BOX25 = [(0, 0, 0.3), (1, 0, 0.2), (0, 1, 0.2), (-1, 0, 0.2), (0, -1, 0.2), (1, 1, 0.1), (1, -1, 0.1), (-1, 1, 0.1), (-1, -1, 0.1),
         (2, 0, 0.1), (0, 2, 0.1), (-2, 0, 0.1), (0, -2, 0.1), (1, 2, 0.05), (-1, 2, 0.05), (1, -2, 0.05), (-1, -2, 0.05),
         (-2, 1, 0.05), (2, -1, 0.05), (-2, -1, 0.05), (2, 1, 0.05), (2, 2, 0.02), (2, -2, 0.02), (-2, 2, 0.02), (-2, -2, 0.02)]
# non-"nice" coefficients: pins the 6-significant-digit rounding of emitted literals
ODD3 = [(0, 0, 0, 0.31234567), (1, 0, 0, 0.1987654321), (-1, 0, 0, 0.0123456789), (0, 1, 0, 0.15151515), (0, -1, 0, 0.1010101),
        (0, 0, 1, 0.0777777), (0, 0, -1, 0.1333333)]
ODD2 = [(0, 0, 0.31234567), (0, 1, 0.1987654321), (1, 0, 0.0123456789), (0, -1, 0.15151515), (-1, 0, 0.1010101), (1, 1, -0.0777777)]

# name, ndim, dims(L,M,N), iterations, points, extra reference options
CASES = [
    ("tiny3d_s1", 3, (12, 10, 9), 4, STAR3, ["--step", "1"]),
    ("tiny3d_s2", 3, (12, 10, 9), 4, STAR3, ["--step", "2"]),
    ("tiny3d_s3", 3, (16, 14, 15), 4, STAR3, ["--step", "3"]),
    ("tiny3d_it5", 3, (12, 10, 9), 5, STAR3, ["--step", "1"]),
    ("cross3d_s1", 3, (11, 10, 12), 4, CROSS3, ["--step", "1", "--dist", "2"]),
    ("cross3d_s2", 3, (13, 12, 14), 4, CROSS3, ["--step", "2", "--dist", "2"]),
    ("odd3d_s1", 3, (9, 11, 10), 4, ODD3, ["--step", "1"]),
    ("odd3d_s2", 3, (12, 11, 13), 4, ODD3, ["--step", "2"]),
    ("tiny2d_s1", 2, (1, 11, 13), 4, STAR2, ["--step", "1"]),
    ("tiny2d_s2", 2, (1, 11, 13), 4, STAR2, ["--step", "2"]),
    ("tiny2d_s3_it100", 2, (1, 23, 21), 100, STAR2, ["--step", "3"]),
    ("cross2d_s1", 2, (1, 12, 11), 4, CROSS2, ["--step", "1", "--dist", "2"]),
    ("box9_s1", 2, (1, 13, 12), 4, BOX9, ["--step", "1"]),
    ("star9_s1", 2, (1, 14, 15), 4, STAR9, ["--step", "1"]),
    ("cross9_s1", 2, (1, 15, 14), 4, CROSS9, ["--step", "1", "--dist", "2"]),
    ("tiny25_s1", 2, (1, 17, 19), 4, BOX25, ["--step", "1"]),
    ("tiny25_s2", 2, (1, 17, 19), 4, BOX25, ["--step", "2"]),
    ("odd2d_s1", 2, (1, 12, 13), 4, ODD2, ["--step", "1"]),
    ("odd2d_s2", 2, (1, 15, 13), 6, ODD2, ["--step", "2"]),
]


def write_stc(path, ndim, dims, iters, pts, iter_token="iterations"):
    L, M, N = dims
    with open(path, "w") as f:
        if ndim == 3:
            f.write("L %d\n" % L)
        f.write("M %d\nN %d\n\n%s %d\n\nstencil\n" % (M, N, iter_token, iters))
        for p in pts:
            f.write(" ".join(repr(v) if isinstance(v, float) else str(v) for v in p) + "\n")


def run_ref(cwd, args):
    try:
        p = subprocess.run([REFBIN] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=10)
    except subprocess.TimeoutExpired:
        return "hang", ""   # the reference spins forever (drstencil.hpp:60: fail state never reaches eof)
    rc = p.returncode if p.returncode >= 0 else 256 + p.returncode
    return rc, p.stdout


def lift(cu_text, name):
    """Pull macros, gold guard+statement, loop increment out of the emitted text."""
    macros = {}
    for m in re.finditer(r"^#define (L|M|N|Iterations|Range|Halo|Dist|Bx|By|Sn) (-?\d+)\s*$", cu_text, re.M):
        macros[m.group(1)] = int(m.group(2))
    g = cu_text.index("__global__ void gold_" + name)
    body = cu_text[g:]
    guard = re.search(r"^\s*if \((.*Halo.*)\) \{\s*$", body, re.M)
    st = body.index("out[", guard.end())
    en = body.index(";", st)
    stmt = body[st:en + 1]
    inc = int(re.search(r"for \(int t = 0; t < Iterations; t \+= (\d+)\)", cu_text).group(1))
    terms = []
    for m in re.finditer(r"\(([^()]+)\) \* in\[([^\]]*)\]\[([^\]]*)\](?:\[([^\]]*)\])?", stmt):
        idx = [m.group(2), m.group(3)] + ([m.group(4)] if m.group(4) is not None else [])
        offs = [int(re.sub(r"^[kji]\+?", "", s) or 0) for s in idx]
        terms.append({"coef": m.group(1), "off": offs})
    return macros, guard.group(1), stmt, inc, terms


DRIVER = r"""
#include <cstdio>
#include <cstdlib>
#include "%(ref)s/common.hpp"
%(defs)s
static void dump(const char* fn, const double* p, size_t n) { FILE* f = fopen(fn, "wb"); fwrite(p, sizeof(double), n, f); fclose(f); }
#if NDIM == 3
static void gold(double* d_in, double* d_out) {
  double (*in)[M][N] = (double (*)[M][N]) d_in;
  double (*out)[M][N] = (double (*)[M][N]) d_out;
  for (int k = 0; k < L; k++) for (int j = 0; j < M; j++) for (int i = 0; i < N; i++)
    if (%(guard)s) { %(stmt)s }
}
#else
static void gold(double* d_in, double* d_out) {
  double (*in)[N] = (double (*)[N]) d_in;
  double (*out)[N] = (double (*)[N]) d_out;
  for (int j = 0; j < M; j++) for (int i = 0; i < N; i++)
    if (%(guard)s) { %(stmt)s }
}
#endif
int main() {
#if NDIM == 3
  size_t n = (size_t)L * M * N;
  double* a = getRandom3DArray(L, M, N);
  double* b = getZero3DArray(L, M, N);
#else
  size_t n = (size_t)M * N;
  double* a = getRandom2DArray(M, N);
  double* b = getZero2DArray(M, N);
#endif
  dump("a0.bin", a, n);
  int launches = 0;
  for (int t = 0; t < Iterations; t += %(inc)d) { gold(a, b); gold(b, a); launches += 2; }
  dump("a.bin", a, n); dump("b.bin", b, n);
  printf("%%d\n", launches);
  return 0;
}
"""


def make_case(tmp, name, ndim, dims, iters, pts, opts):
    stc = "g.stc"
    write_stc(os.path.join(tmp, stc), ndim, dims, iters, pts)
    args = (["--3d"] if ndim == 3 else []) + opts + ["--check", "-o", "g.cu", stc]
    rc, out = run_ref(tmp, args)
    assert rc == 0 and os.path.exists(os.path.join(tmp, "g.cu")), (name, rc, out)
    cu = open(os.path.join(tmp, "g.cu")).read()
    macros, guard, stmt, inc, terms = lift(cu, "g")
    defs = "#define NDIM %d\n" % ndim + "".join("#define %s %d\n" % kv for kv in macros.items() if kv[0] not in ("Bx", "By", "Sn"))
    if ndim == 2:
        defs += "#define L 1\n" if "L" not in macros else ""
    src = DRIVER % dict(ref=REF, defs=defs, guard=guard, stmt=stmt, inc=inc)
    with open(os.path.join(tmp, "drv.cpp"), "w") as f:
        f.write(src)
    subprocess.check_call(["g++", "-O0", "-std=c++17", "-o", "drv", "drv.cpp"], cwd=tmp)
    launches = int(subprocess.check_output(["./drv"], cwd=tmp, text=True).strip())
    shape = dims if ndim == 3 else dims[1:]
    arr = {k: np.fromfile(os.path.join(tmp, k + ".bin"), dtype=np.float64).reshape(shape) for k in ("a0", "a", "b")}
    step = int(opts[opts.index("--step") + 1])
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), a0=arr["a0"], a=arr["a"], b=arr["b"],
                        meta=np.array(json.dumps(dict(ndim=ndim, dims=list(dims), iterations=iters, step=step, launches=launches,
                                                      macros=macros, terms=terms, points=[list(p) for p in pts]))))
    os.remove(os.path.join(tmp, "g.cu"))
    return dict(name=name, macros=macros, launches=launches, nterms=len(terms),
                sum_a=float(arr["a"].sum()), sum_b=float(arr["b"].sum()))


def cli_cases(tmp):
    """Reference CLI behaviour (stdout + exit code + macros) for option/edge cases."""
    write_stc(os.path.join(tmp, "s3.stc"), 3, (512, 512, 512), 4, STAR3)
    write_stc(os.path.join(tmp, "c3.stc"), 3, (512, 512, 512), 4, CROSS3)
    write_stc(os.path.join(tmp, "s2.stc"), 2, (1, 8192, 8192), 4, STAR2)
    write_stc(os.path.join(tmp, "b25.stc"), 2, (1, 8192, 8192), 4, BOX25)
    write_stc(os.path.join(tmp, "typo.stc"), 2, (1, 64, 64), 4, CROSS9, iter_token="iteratioins")
    runs = [
        [], ["--help"], ["-h"], ["s3.stc"], ["--3d", "s3.stc"], ["--3d", "missing.stc"], ["missing.stc"],
        ["--3d", "--step", "s3.stc"], ["--3d", "--bogus", "s3.stc"], ["--3d", "-o", "s3.stc"],
        ["--3d", "c3.stc"], ["--3d", "--dist", "2", "c3.stc"], ["--3d", "--step", "2", "--dist", "2", "c3.stc"],
        ["--3d", "--step", "2", "s3.stc"], ["--3d", "--step", "2", "--dist", "1", "s3.stc"], ["--3d", "--step", "3", "s3.stc"],
        ["--3d", "--step", "2", "--bx", "4", "--by", "4", "s3.stc"], ["--3d", "--step", "3", "--bx", "4", "--merge-forward", "0", "s3.stc"],
        ["--3d", "--gold", "--check", "--prefetch", "--streaming", "s3.stc"],
        ["--3d", "--bx", "32", "--by", "8", "--sn", "64", "--stream-unroll", "8", "--block-merge-x", "4", "--cyclic-merge-y", "2", "s3.stc"],
        ["s2.stc"], ["--step", "2", "s2.stc"], ["--streaming", "s2.stc"], ["--streaming", "--step", "2", "--bx", "4", "s2.stc"],
        ["b25.stc"], ["--step", "2", "b25.stc"], ["--step", "2", "--bx", "8", "b25.stc"], ["--step", "2", "--by", "8", "b25.stc"],
        ["--merge-forward", "100", "b25.stc"], ["--dist", "1", "b25.stc"], ["typo.stc"], ["--dist", "2", "typo.stc"],
        ["--3d", "s2.stc"], ["c3.stc"],
    ]
    out = []
    for args in runs:
        if os.path.exists(os.path.join(tmp, "out.cu")):
            os.remove(os.path.join(tmp, "out.cu"))
        rc, so = run_ref(tmp, args)
        print(args, rc, flush=True)
        rec = dict(args=args, rc=rc, stdout=so if len(so) < 400 else so[:120] + "...<help text>", emitted=False)
        p = os.path.join(tmp, "out.cu")
        if os.path.exists(p):
            cu = open(p).read()
            rec["emitted"] = True
            rec["macros"] = {m.group(1): int(m.group(2)) for m in
                             re.finditer(r"^#define (L|M|N|Iterations|Range|Halo|Dist|Bx|By|Sn) (-?\d+)\s*$", cu, re.M)}
            if "typo.stc" in args:
                rec["macros"].pop("Iterations", None)  # uninitialised in the reference
            if "s2.stc" in args and "--3d" in args or ("c3.stc" in args and "--3d" not in args):
                rec["macros"] = {k: v for k, v in rec["macros"].items() if k in ("Bx", "By", "Sn")}
            km = re.search(r"__global__ void (dr_\w+) \(double \*d_in, double \*d_out\)", cu)
            rec["kernel"] = km.group(1) if km else None
            rec["has_gold"] = "__global__ void gold_" in cu
        out.append(rec)
    return out


def main():
    if not os.path.exists(REFBIN):
        subprocess.check_call(["make", "-C", HERE, "ref"])
    os.makedirs(GOLD, exist_ok=True)
    summary = []
    for case in CASES:
        with tempfile.TemporaryDirectory() as tmp:
            summary.append(make_case(tmp, *case))
            print(summary[-1], flush=True)
    with tempfile.TemporaryDirectory() as tmp:
        cli = cli_cases(tmp)
    with open(os.path.join(GOLD, "ref_cli.json"), "w") as f:
        json.dump(cli, f, indent=1)
    with open(os.path.join(GOLD, "summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    print("wrote", len(summary), "cases and", len(cli), "cli records to", GOLD)


if __name__ == "__main__":
    sys.exit(main())
