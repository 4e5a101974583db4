# one-GPU rehearsals of bench.py --gpus N with the round-4 line (trial of both exchange modes, exchange_ab, per-rank reports with the launch timeline)
set -o pipefail
DRS_REHEARSE=3/8 timeout -k 10 300 python3 bench.py --gpus 8 --steps 20 --warmup 3 > gpurun_out/r04_bench_rehearse_3of8_torch.json 2> gpurun_out/r04_bench_rehearse_3of8_torch.err || echo "rc=$? 3/8 torch"
DRS_REHEARSE=1/4 timeout -k 10 300 python3 bench.py --gpus 4 --steps 20 --warmup 3 > gpurun_out/r04_bench_rehearse_1of4_torch.json 2> gpurun_out/r04_bench_rehearse_1of4_torch.err || echo "rc=$? 1/4 torch"
DRS_REHEARSE=3/8 timeout -k 10 300 python3 bench.py --gpus 8 --steps 20 --warmup 3 --slab-runtime native > gpurun_out/r04_bench_rehearse_3of8_native.json 2> gpurun_out/r04_bench_rehearse_3of8_native.err || echo "rc=$? 3/8 native"
DRS_BENCH_BACKEND=gloo DRS_BENCH_ONE_GPU=1 timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r04_bench_two_rank_processes_one_gpu_gloo.json 2> gpurun_out/r04_bench_two_rank_processes_one_gpu_gloo.err || echo "rc=$? gloo 2"
for f in rehearse_3of8_torch rehearse_1of4_torch rehearse_3of8_native two_rank_processes_one_gpu_gloo; do
python3 -c "
import json
d=json.load(open('gpurun_out/r04_bench_$f.json'))
r=(d.get('ranks') or [{}])[0]
print('$f', round(d['value'],1), 'eff', d.get('efficiency_vs_n1'), 'verified', d.get('verified'), 'ab', d.get('exchange_ab'), 'calib', {k:v for k,v in (d['config'].get('exchange_calibration') or {}).items() if k in ('model_every','chosen_every','trial_ms_per_step','ranks_agreed')}, 'timeline', r.get('launch_timeline_us'), 'rccl', r.get('rccl'))
" || echo "no line $f"
done
