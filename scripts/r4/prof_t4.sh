set -o pipefail
P="fetch write tcc sq sq2 grbm"
PASSES="$P" bash scripts/profile_bench.sh r04h_c4f64_temporal4 --workload c4f64 --kernel-args "--3d --dtype fp64 --step 4 --temporal 1 --skew 1 --pin 1 --exact-y 1 --prefetch --prefetch-depth 2 --bx 68 --by 11 --block-merge-x 2 --block-merge-y 2 --sn 256 --xcd-remap 4" || echo "rc=$? c4f64 t4"
PASSES="$P" bash scripts/profile_bench.sh r04h_c3f64_temporal4 --workload c3f64 --kernel-args "--3d --dtype fp64 --step 4 --temporal 1 --skew 1 --pin 1 --exact-y 1 --prefetch --prefetch-depth 2 --bx 68 --by 11 --block-merge-x 2 --block-merge-y 2 --sn 64 --xcd-remap 4" || echo "rc=$? c3f64 t4"
