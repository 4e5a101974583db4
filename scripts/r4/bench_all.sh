# every workload's bench.py line in one call (= one box), then the one-GPU rehearsals of the N > 1 path
set -o pipefail
mkdir -p gpurun_out
for w in c4 c3 c2 c5 c2f64 c3f64 c4f64 s_2d5pt_star s_2d5pt_cross s_2d9pt_box s_2d9pt_star s_2d9pt_cross s_2d25pt_box s_3d9pt_cross; do
  timeout -k 10 400 python3 bench.py --workload $w > gpurun_out/r04_bench_$w.json 2> gpurun_out/r04_bench_$w.err || echo "bench $w rc=$?"
  python3 -c "
import json,sys
d=json.load(open('gpurun_out/r04_bench_$w.json'))
t3=d.get('temporal_step3_kernel') or {}
print('$w', round(d['value'],1), 'frac', round(d['roofline']['frac'],4), 'traffic', d['roofline']['traffic'] and round(d['roofline']['traffic']/d['roofline']['algorithmic_bytes_per_launch'],3), 'verified', d['verified'], [s_['position'] for s_ in (d['verification']['vs_cpu_oracle_slab'] or {}).get('slabs',[])], 't3', t3.get('GStencil_per_s') and round(t3['GStencil_per_s'],1), t3.get('verified'))
" || echo "no line for $w"
done
