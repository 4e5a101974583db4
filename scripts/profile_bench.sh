#!/bin/bash
# profile_bench.sh <tag> [bench.py args...] -- rocprofv3 passes over one bench.py command on the GPU box:
# a kernel trace with --stats, then one --pmc pass per counter group (never combined with other trace domains),
# then scripts/summarize_pmc.py -> gpurun_out/<tag>_counters.json + <tag>_kernel_stats.csv.
# Kernels are built BEFORE the profiled runs (bench.py --prebuild-only) and the runs carry DRS_NO_COMPILE=1: under
# `rocprofv3 --pmc` the GPU is initialised before main(), so the profiled process must never start hipcc.
# PASSES (env): space-separated list of pass names to run besides trace; default "fetch write tcc sq".
set -o pipefail
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
STEPS=${STEPS:-10}
B=(python3 bench.py --steps $STEPS --warmup 3 --no-cpu-baseline --no-verify --headline-only "$@")
out=gpurun_out/prof_$tag
mkdir -p $out
python3 bench.py --prebuild-only --headline-only "$@" > $out/prebuild.log 2>&1 || exit 9
export DRS_NO_COMPILE=1
declare -A PMC=(
  [fetch]="FETCH_SIZE"
  [write]="WRITE_SIZE"
  [tcc]="TCC_HIT_sum TCC_MISS_sum"
  [sq]="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
  [sq2]="SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"
  [sq3]="SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAVE_CYCLES SQ_IFETCH SQ_INSTS_BRANCH SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL"
  [ea]="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum"
  [eastall]="TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum"
  [grbm]="GRBM_GUI_ACTIVE GRBM_COUNT"
)
timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- "${B[@]}" > $out/trace.log 2>&1 || exit 2
# a failed pass leaves a partial (or a previous run's) counter directory behind: it is removed and named to summarize_pmc.py, which
# then writes no traffic figure when fetch or write is missing, and the script exits non-zero after summarising what it has
failed=""
for pass in ${PASSES:-fetch write tcc sq}; do
  rm -rf $out/$pass
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc ${PMC[$pass]} --output-format csv -d $out/$pass -- "${B[@]}" > $out/$pass.log 2>&1 || { echo "pass $pass failed"; tail -3 $out/$pass.log; rm -rf $out/$pass; failed="$failed $pass"; }
done
FAILED_PASSES="$failed" python3 scripts/summarize_pmc.py $tag "${B[*]}" $STEPS || exit 7
[ -z "$failed" ] || { echo "profile_bench.sh: failed passes:$failed"; exit 6; }
