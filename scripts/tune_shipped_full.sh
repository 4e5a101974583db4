#!/bin/bash
# Exhaustive tuner runs over the shipped 2D stencils (reference sizes 8192^2, fp64 like the reference), steps 1 and 2.
# usage: tune_shipped_full.sh <outdir> <per-stencil budget in s> <stencil>...
out=$1; budget=$2; shift 2
mkdir -p $out
for s in "$@"; do
  ord=1; case $s in 2d9pt_star|2d25pt_box) ord=2;; esac
  d3=""; case $s in 3d*) d3="--3d";; esac
  steps=${STEPS:-1,2}
  timeout -k 10 $((budget + 90)) python drstencil_amd/tuner/tuning.py benchmarks/$s/$s.stc $d3 --dtype fp64 --order $ord --steps $steps --seed 5 --budget $budget --out $out/${s}_s${steps/,/} > $out/${s}_s${steps/,/}.log 2>&1
  echo "== $s steps $steps"; tail -4 $out/${s}_s${steps/,/}.log | head -3
done
