# times every compiled variant of scripts/r4/micro/pipeline_variants.py (80 launches each after 10 warm-up launches)
for v in base no_second_barrier no_barriers no_tail_writes no_rim_reads no_neighbour_reads no_writes_no_barriers no_lds_at_all no_lds_keep_barriers no_reads_no_rim base; do
  printf "%-26s " $v; timeout -k 10 120 scripts/r4/micro/variants/$v | grep "computation time\|GStencil" | tr '\n' ' '; echo
done
