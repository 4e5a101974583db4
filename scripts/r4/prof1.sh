set -o pipefail
P="fetch write tcc sq sq2 grbm"
PASSES="$P" bash scripts/profile_bench.sh r04_c5_tile128x32 --workload c5 --kernel-args "--dtype fp64 --bx 64 --by 4 --block-merge-x 2 --block-merge-y 8 --xcd-remap 0 --order rows" || echo "rc=$?"
PASSES="$P" bash scripts/profile_bench.sh r04_c5_tile128x64 --workload c5 --kernel-args "--dtype fp64 --bx 64 --by 8 --block-merge-x 2 --block-merge-y 8 --xcd-remap 0 --order rows" || echo "rc=$?"
PASSES="$P" bash scripts/profile_bench.sh r04_c3_sn64 --workload c3 --kernel-args "--3d --dtype fp32 --step 2 --prefetch --prefetch-depth 3 --bx 32 --by 16 --block-merge-x 4 --block-merge-y 2 --sn 64 --xcd-remap 2 --cc-opt -fno-slp-vectorize" || echo "rc=$?"
PASSES="$P" bash scripts/profile_bench.sh r04_s_2d25pt_box_t2 --workload s_2d25pt_box || echo "rc=$?"
