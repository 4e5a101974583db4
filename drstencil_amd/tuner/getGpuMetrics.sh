#!/bin/bash
# getGpuMetrics.sh -- counterpart of the reference's benchmarks/<stencil>/getGpuMetrics.sh:1-5: one gpuMetrics.csv row per profiled
# configuration.  Ours keeps one directory prof/<name>/ (three rocprofv3 runs) plus prof/<name>.log per configuration; the CSV header
# is written by getGpuMetrics.py itself (the reference copies a header-only template).
here=$(cd "$(dirname "$0")" && pwd)
rm -f gpuMetrics.csv
for log in prof/*.log; do
    [ -e "$log" ] || continue
    name=$(basename "${log%.log}")
    python "$here/getGpuMetrics.py" "$name"
done
