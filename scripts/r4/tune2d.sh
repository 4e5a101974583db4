# the tuner's round-4 space (taps / rows / pin emissions, six rows per lane, skewed row pipelines) over the shipped 2D specs and C5, each
# with --write-defaults into a scratch table: tuner -> table rows for real.  DRS_TUNE_BUILD_ONLY=1: compile only (no GPU).
set -o pipefail
OUT=${1:-gpurun_out/tune2d}
mkdir -p $OUT
run() {  # name stc order steps
  timeout -k 10 ${TUNE_TIMEOUT:-280} python3 -m drstencil_amd.tuner.tuning $2 --dtype fp64 --order $3 --steps $4 --space r4 --emit taps,rows,pin --budget 240 --jobs 8 \
      --out $OUT/$1 --write-defaults --defaults-table $OUT/tuned_defaults.tsv > $OUT/$1.txt 2>&1 || echo "rc=$? $1"
  grep -A3 "^best" $OUT/$1.txt | cut -c1-260; grep "timed," $OUT/$1.txt
}
run 2d5pt_star  benchmarks/2d5pt_star/2d5pt_star.stc   1 2
run 2d5pt_cross benchmarks/2d5pt_cross/2d5pt_cross.stc 1 2
run 2d9pt_box   benchmarks/2d9pt_box/2d9pt_box.stc     1 2
run 2d9pt_star  benchmarks/2d9pt_star/2d9pt_star.stc   2 2
run 2d9pt_cross benchmarks/2d9pt_cross/2d9pt_cross.stc 2 2
run c5_2d25pt_box benchmarks/configs/c5_2d25pt_box_16384.stc 2 1
