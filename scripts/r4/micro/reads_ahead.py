#!/usr/bin/env python3
"""Timing experiment: the 4-stage pipeline with every stage's LDS reads issued one stage EARLIER (before the previous stage's FMAs), by
re-ordering the emitted source text.  Legal: between the two barriers of an iteration no slot is written.  The program is emitted with
--check, so its output says whether the result still equals the gold kernel's.  usage: reads_ahead.py [prefetch depth]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
OUT = os.path.join(ROOT, "scripts", "r4", "micro", "variants")
os.makedirs(OUT, exist_ok=True)
pd = sys.argv[1] if len(sys.argv) > 1 else "2"
OPTS = ("--3d --dtype fp64 --step 4 --temporal 1 --skew 1 --pin 1 --exact-y 1 --prefetch --prefetch-depth %s --bx 68 --by 11 --block-merge-x 2 --block-merge-y 2 --sn 256 --xcd-remap 4 --check" % pd).split()
base = os.path.join(OUT, "ra_base_pd%s.hip" % pd)
subprocess.check_call([os.path.join(ROOT, "bin", "drstencil")] + OPTS + ["-o", base, os.path.join(ROOT, "benchmarks", "configs", "c4_3d7pt_star_1024.stc")], stdout=subprocess.DEVNULL)
src = open(base).read()


def reorder(text):
    lines = text.split("\n")
    out = []
    i = 0
    n_blocks = 0
    while i < len(lines):
        if lines[i].strip() == "// stage 3":
            # collect the four stage segments up to the second barrier of this iteration
            j = i
            segs = []
            while "every stage has read its slot" not in lines[j]:
                if re.match(r"\s*// stage \d$", lines[j]):
                    segs.append([])
                segs[-1].append(lines[j])
                j += 1
            full = len(segs) == 4 and all(any("__builtin_fma" in l for l in seg) and any("xedge_r" in l for l in seg) for seg in segs)
            if full:
                reads, rest = [], []
                for seg in segs:
                    # the reads part: up to and including the closing brace of the `if (xedge_r) { ... }` block
                    k = max(idx for idx, l in enumerate(seg) if l.strip() == "}" and any("xedge_r" in x for x in seg[:idx]))
                    first_fma = next(idx for idx, l in enumerate(seg) if "__builtin_fma" in l)
                    k = min(k, first_fma - 1)
                    reads.append(seg[:k + 1]); rest.append(seg[k + 1:])
                out += reads[0] + reads[1] + rest[0] + reads[2] + rest[1] + reads[3] + rest[2] + rest[3]
                n_blocks += 1
            else:
                for seg in segs:
                    out += seg
            i = j
            continue
        out.append(lines[i])
        i += 1
    return "\n".join(out), n_blocks


new, nb = reorder(src)
print("re-ordered", nb, "iteration bodies")
src = re.sub(r"#define Iterations \d+", "#define Iterations 160", src)
new = re.sub(r"#define Iterations \d+", "#define Iterations 160", new)
procs = []
for name, text in (("ra_base_pd%s" % pd, src), ("ra_ahead_pd%s" % pd, new)):
    p = os.path.join(OUT, name + ".hip")
    open(p, "w").write(text)
    procs.append((name, subprocess.Popen(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "drstencil_amd", "csrc", "support"),
                                          "-Rpass-analysis=kernel-resource-usage", "-o", os.path.join(OUT, name), p], stderr=subprocess.PIPE, text=True)))
for name, pr in procs:
    err = pr.communicate()[1]
    blk = err[err.find("Function Name: dr_"):]
    g = lambda k: (re.search(k + r":\s*(\d+)", blk) or [None, "?"])[1]
    print("%-20s rc=%d vgprs %s scratch %s" % (name, pr.returncode, g("VGPRs"), g(r"ScratchSize \[bytes/lane\]")))
