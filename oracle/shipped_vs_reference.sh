#!/bin/bash
# TEST INFRASTRUCTURE (authoring container only: runs oracle/_ref/drstencil_ref, the reference generator built by oracle/Makefile).
# The shipped specs (benchmarks/<name>/<name>.stc -- the ones the tuned-defaults table has rows for) x reference-style command lines through the
# reference binary and bin/drstencil: exit code and stdout must be identical byte for byte (our remarks go to stderr).  Companion of
# oracle/fuzz_vs_reference.py, whose random shapes never hit the table.  (Not in the list: `--step 0`, which the reference accepts -- it emits a
# program whose time loop never ends, `t += 2 * step` -- and this generator refuses with "Illegal input.", exit code 255: DESIGN.md section 5.)
HERE=$(cd "$(dirname "$0")" && pwd); REPO=$(dirname "$HERE"); T=$(mktemp -d); n=0; bad=0
for s in 2d5pt_star 2d5pt_cross 2d9pt_box 2d9pt_star 2d9pt_cross 2d25pt_box 3d7pt_star 3d9pt_cross; do
  d3=""; [ ${s:0:2} = 3d ] && d3="--3d"
  for opts in "" "--step 2" "--step 3" "--step 4" "--step 2 --dist 2" "--step 2 --dist 4" "--dist 9" "--streaming" "--step 2 --streaming" "--prefetch" "--step 2 --bx 64 --by 4" \
              "--step 2 --merge-forward 3" "--block-merge-x 2 --cyclic-merge-y 2" "--step 2 --check" "--bx 7"; do
    "$HERE/_ref/drstencil_ref" $d3 $opts -o $T/a.cu "$REPO/benchmarks/$s/$s.stc" > $T/a.out 2>/dev/null; ra=$?
    "$REPO/bin/drstencil" $d3 $opts -o $T/b.hip "$REPO/benchmarks/$s/$s.stc" > $T/b.out 2>/dev/null; rb=$?
    n=$((n+1))
    if [ $ra -ne $rb ] || ! cmp -s $T/a.out $T/b.out; then bad=$((bad+1)); echo "DIFFERENCE $s [$opts] exit codes $ra / $rb"; fi
  done
done
rm -rf $T
echo "$n command lines on 8 shipped specs against the reference binary: $bad differences in exit code or stdout"
[ $bad -eq 0 ]
