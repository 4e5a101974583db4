#!/bin/bash
# probe_slow_box.sh -- the fused step-3 kernel takes 1.55-1.68 ms on most MI355X devices and ~1.9 ms on some.  Time it on
# the device at hand; if this is one of the slow ones, collect the counters (clock from GRBM_GUI_ACTIVE, VALU activity, occupancy, HBM traffic)
# of it and of the step-2 headline for comparison with profiles/r03_s3_64x8_sn64_counters.json (a fast device).
S3="--3d --dtype fp32 --step 3 --prefetch --prefetch-depth 1 --block-merge-x 4 --order rows --pack 0 --cc-opt -fno-slp-vectorize --bx 64 --by 8 --block-merge-y 2 --sn 64 --xcd-remap 2"
ms=$(python3 bench.py --headline-only --no-cpu-baseline --no-verify --kernel-args "$S3" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f' % d['roofline']['avg_launch_ms'])")
ms2=$(python3 bench.py --headline-only --no-cpu-baseline --no-verify 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f' % d['roofline']['avg_launch_ms'])")
echo "this device: fused step 3 $ms ms per launch, step-2 headline $ms2 ms"
slow=$(python3 -c "print(1 if float('$ms') > ${SLOW_MS:-1.75} else 0)")
if [ "$slow" = "1" ] || [ -n "$FORCE" ]; then
  PASSES="fetch write tcc sq sq2 grbm" bash scripts/profile_bench.sh r03_slowbox_s3 --kernel-args "$S3" | tail -1
  PASSES="sq sq2 grbm" bash scripts/profile_bench.sh r03_slowbox_headline | tail -1
else
  echo "a fast device: nothing collected"
fi
