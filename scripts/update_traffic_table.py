#!/usr/bin/env python3
"""update_traffic_table.py <workload>=<tag> ... -- fold the FETCH/WRITE counters of scripts/profile_bench.sh runs (gpurun_out/<tag>_counters.json)
into profiles/traffic_by_options.json, the table bench.py's `roofline.traffic` is read from (keyed by workload and generator options), and copy
the per-run summaries to profiles/<tag>_counters.json / <tag>_kernel_stats.csv.  Refuses a summary without both counters (a failed --pmc pass)."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
table_path = os.path.join(ROOT, "profiles", "traffic_by_options.json")
table = json.load(open(table_path))
for arg in sys.argv[1:]:
    workload, tag = arg.split("=")
    src = os.path.join(ROOT, "gpurun_out", tag + "_counters.json")
    c = json.load(open(src))
    if "traffic_bytes_per_launch" not in c or c.get("failed_passes"):
        print("%s: no traffic figure in %s (failed passes: %s): table left alone" % (workload, src, c.get("failed_passes")))
        continue
    table.setdefault(workload, {})[c["generator_options"]] = {
        "traffic_bytes_per_launch": c["traffic_bytes_per_launch"], "fetch_size_kib": c["FETCH_SIZE"], "write_size_kib": c["WRITE_SIZE"],
        "traffic_over_algorithmic": c.get("traffic_over_algorithmic"), "l2_hit_rate": c.get("l2_hit_rate"),
        "source": "profiles/%s_counters.json (scripts/profile_bench.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, timed dispatches only; FETCH_SIZE x2 per MI355X_MICROARCH.md)" % tag}
    shutil.copy(src, os.path.join(ROOT, "profiles", tag + "_counters.json"))
    ks = os.path.join(ROOT, "gpurun_out", tag + "_kernel_stats.csv")
    if os.path.exists(ks):
        shutil.copy(ks, os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv"))
    print("%s: %s -> %.3f GB per launch (%.3fx algorithmic), timed avg %.4f ms" % (workload, tag, c["traffic_bytes_per_launch"] / 1e9, c.get("traffic_over_algorithmic", 0), c.get("timed_avg_ns", 0) / 1e6))
json.dump(table, open(table_path, "w"), indent=1)
