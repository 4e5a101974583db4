#!/usr/bin/env python3
"""bench_table.py [tag] -- the roofline table of DESIGN.md section 3.2 from the committed bench.py lines profiles/<tag>_bench_<workload>.json
(default tag r04): replaces the text between the BENCH-TABLE markers."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
order = ["c4", "c3", "c2", "c5", "c2f64", "c3f64", "c4f64", "s_2d5pt_star", "s_2d5pt_cross", "s_2d9pt_box", "s_2d9pt_star", "s_2d9pt_cross", "s_2d25pt_box", "s_3d9pt_cross"]
rows = ["| workload | headline kernel (generator options after the problem's) | launch | GStencil/s | frac of 8 TB/s | traffic / algorithmic | side measurements (same run, each checked against its gold kernel on the whole grid) |", "|---|---|---|---|---|---|---|"]
for w in order:
    p = os.path.join(ROOT, "profiles", "%s_bench_%s.json" % (tag, w))
    if not os.path.exists(p):
        continue
    d = json.load(open(p))
    r = d["roofline"]
    opts = d["config"]["generator_options"]
    opts = re.sub(r"^(--3d )?--dtype fp(32|64) ?", "", opts)
    side = []
    for key, label in (("step1_kernel", "step 1"), ("fused_multistep_kernel", "fused step %s"), ("temporal_step2_kernel", "temporal 2 stages"), ("temporal_step3_kernel", "**temporal 3 stages**"), ("temporal_step4_kernel", "**temporal 4 stages**")):
        s = d.get(key)
        if s:
            lab = label % s["step"] if "%s" in label else label
            side.append("%s: %.0f (%.3f)" % (lab, s["GStencil_per_s"], s["roofline_frac"]))
    tr = r["traffic"] / r["algorithmic_bytes_per_launch"] if r.get("traffic") else None
    rows.append("| %s | `%s` | %.4f ms | **%.0f** | **%.3f** | %s | %s |" % (d["config"]["workload"].split(",")[0], opts, r["avg_launch_ms"], d["value"], r["frac"],
                                                                  "%.3f×" % tr if tr else "—", "; ".join(side) or "—"))
text = "\n".join(rows)
path = os.path.join(ROOT, "DESIGN.md")
s = open(path).read()
a = s.index("<!-- BENCH-TABLE -->")
b = s.find("<!-- /BENCH-TABLE -->")
if b < 0:
    s = s[:a] + "<!-- BENCH-TABLE -->\n" + text + "\n<!-- /BENCH-TABLE -->" + s[a + len("<!-- BENCH-TABLE -->"):]
else:
    s = s[:a] + "<!-- BENCH-TABLE -->\n" + text + "\n" + s[b:]
open(path, "w").write(s)
print(text)
