#!/bin/bash
# Time-bounded tuner runs over the eight shipped stencils (reference sizes, fp64 like the reference), steps 1 and 2.
out=${1:-gpurun_out/tune_shipped}
mkdir -p $out
for s in 2d5pt_star 2d5pt_cross 2d9pt_box 2d9pt_star 2d25pt_box; do
  ord=1; case $s in 2d9pt_star|2d25pt_box) ord=2;; esac
  timeout -k 10 150 python drstencil_amd/tuner/tuning.py benchmarks/$s/$s.stc --dtype fp64 --order $ord --steps 1,2 --max-configs 48 --seed 7 --out $out/$s > $out/$s.log 2>&1
  tail -4 $out/$s.log | head -3
done
for s in 3d7pt_star 3d9pt_cross; do
  timeout -k 10 200 python drstencil_amd/tuner/tuning.py benchmarks/$s/$s.stc --3d --dtype fp64 --order 1 --steps 1,2 --max-configs 64 --seed 7 --out $out/$s > $out/$s.log 2>&1
  tail -4 $out/$s.log | head -3
done
