#!/usr/bin/env python3
"""Timing experiment on the emitted 4-stage pipeline (NOT a product path: the variants compute wrong results).  Takes the standalone
program the generator emits for the c4f64 4-stage row, deletes one ingredient at a time from the source text (second barrier, both
barriers, the LDS writes of the tail, the conditional rim reads, the neighbour reads) and compiles each variant; run.sh times them on the
GPU.  What a deletion saves is what that ingredient costs in the lock-step of the 748-lane workgroup."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
OUT = os.path.join(ROOT, "scripts", "r4", "micro", "variants")
os.makedirs(OUT, exist_ok=True)
OPTS = "--3d --dtype fp64 --step 4 --temporal 1 --skew 1 --pin 1 --exact-y 1 --prefetch --prefetch-depth 2 --bx 68 --by 11 --block-merge-x 2 --block-merge-y 2 --sn 256 --xcd-remap 4".split()
src_path = os.path.join(OUT, "base.hip")
subprocess.check_call([os.path.join(ROOT, "bin", "drstencil")] + OPTS + ["-o", src_path, os.path.join(ROOT, "benchmarks", "configs", "c4_3d7pt_star_1024.stc")], stdout=subprocess.DEVNULL)
src = open(src_path).read()
src = re.sub(r"#define Iterations \d+", "#define Iterations 160", src)
assert "#define Iterations 160" in src


def drop_lines(text, pred):
    return "\n".join(l for l in text.split("\n") if not pred(l)) + "\n"


def no_b2(t):
    return drop_lines(t, lambda l: "every stage has read its slot" in l)


def no_bar(t):
    return drop_lines(t, lambda l: "__syncthreads();" in l and ("every stage has read" in l or "arriving planes" in l))


def no_writes(t):
    # the writes stay in the program behind a condition that is never true at run time (the arrays are never that close), or the compiler
    # would drop the whole exchange; the prologue's writes stay as they are
    head, loop = t.split("for (int k = k0; k < k1;", 1)
    loop = re.sub(r"\n(\s*)((?:if \(hdo\d+\) )?\*\(vec_t\*\)&shm\[[^\n]*;)", r"\n\1if (drs_never) { \2 }", loop)
    head = head.replace("\n{\n", "\n{\n    const bool drs_never = ((const char*)d_out - (const char*)d_in) == 24;\n", 1) if "drs_never" not in head else head
    return head + "for (int k = k0; k < k1;" + loop


def no_rim(t):
    return re.sub(r"\n\s*if \(xedge_[lr]\) \{\n[^}]*\}", "", t)


def no_reads(t):
    return re.sub(r"(rv(\d)_\d) = \*\(const vec_t\*\)&shm\[sb \+ \(-?\d+\)\];", r"\1 = o\2_0_0;", t)


def lds_rim(t):      # every lane reads its x neighbours from the slot (no DPP, no branch)
    return t


VARIANTS = {
    "base": lambda t: t,
    "no_second_barrier": no_b2,
    "no_barriers": no_bar,
    "no_tail_writes": no_writes,
    "no_rim_reads": no_rim,
    "no_neighbour_reads": no_reads,
    "no_lds_at_all": lambda t: no_reads(no_rim(no_writes(no_bar(t)))),
    "no_writes_no_barriers": lambda t: no_writes(no_bar(t)),
    "no_lds_keep_barriers": lambda t: no_reads(no_rim(no_writes(t))),
    "no_reads_no_rim": lambda t: no_reads(no_rim(t)),
}
procs = []
for name, f in VARIANTS.items():
    p = os.path.join(OUT, name + ".hip")
    open(p, "w").write(f(src))
    procs.append((name, subprocess.Popen(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "drstencil_amd", "csrc", "support"),
                                          "-Rpass-analysis=kernel-resource-usage", "-o", os.path.join(OUT, name), p], stderr=subprocess.PIPE, text=True)))
for name, pr in procs:
    err = pr.communicate()[1]
    blk = err[err.find("Function Name: dr_"):]
    g = lambda k: (re.search(k + r":\s*(\d+)", blk) or [None, "?"])[1]
    print("%-24s rc=%d vgprs %s scratch %s lds %s" % (name, pr.returncode, g("VGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"LDS Size \[bytes/block\]")))
