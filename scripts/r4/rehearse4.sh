# four rank PROCESSES of bench.py --gpus 4 on ONE GPU, halos through gloo staged in host memory: the launcher, both middle ranks with two distinct
# neighbours, the exchange choreography and the self-check of a real 4-GPU run; only the RCCL transport and the rate are not real
set -o pipefail
DRS_BENCH_BACKEND=gloo DRS_BENCH_ONE_GPU=1 timeout -k 10 700 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 4 --steps 3 --warmup 1 > gpurun_out/r04_bench_four_rank_processes_one_gpu_gloo.json 2> gpurun_out/r04_bench_four_rank_processes_one_gpu_gloo.err || echo "rc=$? gloo 4"
python3 -c "
import json
d=json.load(open('gpurun_out/r04_bench_four_rank_processes_one_gpu_gloo.json'))
print(round(d['value'],1), 'verified', d.get('verified'), 'ab', d.get('exchange_ab'), 'calib', {k:v for k,v in (d['config'].get('exchange_calibration') or {}).items() if k in ('model_every','chosen_every','trial_ms_per_step','ranks_agreed')})
for r in d.get('ranks') or []: print(r.get('rank'), r.get('device'), r.get('launch_timeline_us'))
" || tail -20 gpurun_out/r04_bench_four_rank_processes_one_gpu_gloo.err
