# 700 random configurations of the tuner's round-4 3D space (taps / rows / pin emissions) for the shipped 3d9pt_cross 512^3 fp64 step 2
set -o pipefail
OUT=${1:-gpurun_out/tune3d_cross}
mkdir -p $OUT
timeout -k 10 ${TUNE_TIMEOUT:-900} python3 -m drstencil_amd.tuner.tuning benchmarks/3d9pt_cross/3d9pt_cross.stc --3d --dtype fp64 --order 2 --steps 2 --space r4 --emit taps,rows,pin --max-configs 700 --seed 7 \
    --budget 800 --jobs 8 --out $OUT --write-defaults --defaults-table $OUT/tuned_defaults.tsv > $OUT.txt 2>&1 || echo "rc=$?"
grep -A6 "^best" $OUT.txt | cut -c1-260; grep "timed,\|configurations," $OUT.txt | cut -c1-200
