#!/usr/bin/env python3
"""Instruction mix of one kernel in a `hipcc --save-temps` assembly file (.s): counts per mnemonic class over the
whole function and over its hottest loop (the largest backward-branch body).  Usage: isa_stats.py file.s dr_<name>"""
import collections
import re
import sys


def body(path, fn):
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith(fn + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    return lines[start:end + 1]


def classify(m):
    if m.startswith("v_pk_fma"): return "v_pk_fma"
    if m.startswith(("v_fma_", "v_fmac_")): return "v_fma/fmac" + ("_dpp" if m.endswith("_dpp") else "")
    if m.startswith("v_pk_mul"): return "v_pk_mul"
    if m.startswith("v_mul_f"): return "v_mul" + ("_dpp" if m.endswith("_dpp") else "")
    if m.startswith("v_mov") and m.endswith("_dpp"): return "v_mov_dpp"
    if m.startswith(("v_mov", "v_pk_mov")): return "v_mov"
    if m.startswith("v_accvgpr"): return "v_accvgpr"
    if m.startswith("v_cndmask"): return "v_cndmask"
    if m.startswith("v_"): return "v_other"
    if m.startswith("ds_read") or m.startswith("ds_load"): return "ds_read"
    if m.startswith("ds_write") or m.startswith("ds_store"): return "ds_write"
    if m.startswith(("global_load", "buffer_load")): return "vmem_load" + ("_lds" if "lds" in m else "")
    if m.startswith(("global_store", "buffer_store")): return "vmem_store"
    if m.startswith("scratch_"): return "scratch"
    if m.startswith("s_waitcnt"): return "s_waitcnt"
    if m.startswith("s_barrier"): return "s_barrier"
    if m.startswith(("s_cbranch", "s_branch")): return "s_branch"
    if m.startswith("s_"): return "s_other"
    return "other"


def stats(lines):
    c = collections.Counter()
    for l in lines:
        t = l.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        c[classify(t.split()[0])] += 1
    return c


def main():
    path, fn = sys.argv[1], sys.argv[2]
    b = body(path, fn)
    # hottest loop: label .LBBx_y ... s_cbranch .LBBx_y with the most instructions in between
    labels = {l.strip()[:-1]: i for i, l in enumerate(b) if re.match(r"^\.LBB\d+_\d+:", l.strip())}
    best = (0, 0, 0)
    for i, l in enumerate(b):
        m = re.match(r"\s*s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"\s*s_branch\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i and i - labels[m.group(1)] > best[0]:
            best = (i - labels[m.group(1)], labels[m.group(1)], i)
    for name, part in (("function", b), ("hottest loop (%d lines)" % best[0], b[best[1]:best[2] + 1])):
        c = stats(part)
        tot = sum(c.values())
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        print("%s: %d instructions, %d VALU" % (name, tot, valu))
        for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
            print("   %-16s %6d  %5.1f %%%s" % (k, v, 100.0 * v / tot, "  (%.1f %% of VALU)" % (100.0 * v / valu) if k.startswith("v_") and valu else ""))


if __name__ == "__main__":
    main()
