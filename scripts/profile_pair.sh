#!/bin/bash
# profile_pair.sh <tagA> "<kernel args A>" <tagB> "<kernel args B>" [workload] -- the PMC passes of scripts/profile_bench.sh for two
# kernel configurations of one workload in ONE call, i.e. on one device (boxes differ by several per cent on these kernels)
wl=${5:-c4}
PASSES="${PASSES:-fetch write tcc sq sq2 sq3 grbm}" bash scripts/profile_bench.sh "$1" --workload $wl --kernel-args "$2" || echo "profile $1 rc=$?"
PASSES="${PASSES:-fetch write tcc sq sq2 sq3 grbm}" bash scripts/profile_bench.sh "$3" --workload $wl --kernel-args "$4" || echo "profile $3 rc=$?"
