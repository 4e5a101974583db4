#!/usr/bin/env python3
"""shipped_table.py <results dir>... -- best configuration per shipped stencil and step from the tuner's results.jsonl files."""
import glob
import json
import os
import sys

rows = []
for d in sys.argv[1:]:
    for f in sorted(glob.glob(os.path.join(d, "*", "results.jsonl"))):
        stencil = os.path.basename(os.path.dirname(f)).rsplit("_s", 1)[0]
        rs = [json.loads(l) for l in open(f)]
        ok = [r for r in rs if r.get("duration_ns")]
        for st in sorted(set(r["step"] for r in ok)):
            sel = sorted([r for r in ok if r["step"] == st], key=lambda r: -r["GStencil"])
            n_all = sum(1 for r in rs if r.get("step", st) == st)
            b = sel[0]
            rows.append((stencil, st, len(sel), b["GStencil"], b["GBps"], b["frac"], b["duration_ns"], b["name"]))
print("| stencil | step | configurations timed | best GStencil/s | achieved GB/s | frac of 8 TB/s | launch | best configuration |")
print("|---|---|---|---|---|---|---|---|")
for r in sorted(rows):
    print("| %s | %d | %d | %.1f | %.0f | %.2f | %.3f ms | `%s` |" % (r[0], r[1], r[2], r[3], r[4], r[5], r[6] / 1e6, r[7]))
