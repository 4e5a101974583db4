#!/bin/bash
# profile_bench.sh <tag> [bench.py args...] -- rocprofv3 passes over one bench.py command on the GPU box:
# a kernel trace with --stats, then one --pmc pass per counter group (never combined with other trace domains),
# then scripts/summarize_pmc.py -> gpurun_out/<tag>_counters.json + <tag>_kernel_stats.csv
set -o pipefail
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
B="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --headline-only $*"
out=gpurun_out/prof_$tag
mkdir -p $out
timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $B > $out/trace.log 2>&1 || exit 2
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $B > $out/fetch.log 2>&1 || exit 3
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- $B > $out/write.log 2>&1 || exit 4
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/tcc -- $B > $out/tcc.log 2>&1 || exit 5
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/sq -- $B > $out/sq.log 2>&1 || exit 6
python3 scripts/summarize_pmc.py $tag "$B"
