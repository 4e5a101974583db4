for v in ra_base_pd2 ra_base_pd1 ra_ahead_pd1 ra_ahead_pd2 ra_base_pd2 ra_ahead_pd1; do
  printf "%-16s " $v; timeout -k 10 200 scripts/r4/micro/variants/$v | grep "computation time\|GStencil\|RMS\|Error\|error" | tr '\n' ' '; echo
done
