set -o pipefail
mkdir -p gpurun_out
S0=$(date +%s)
GPT_BENCH_BACKEND=gloo GPT_BENCH_ONE_GPU=1 GPT_BENCH_WATCHDOG=1000 timeout -k 10 1100 python bench.py --gpus 2 --steps 20 --warmup 3 --repeats 2 > gpurun_out/r3_bench_n2.json 2> gpurun_out/r3_bench_n2.err; rc=$?
echo "rc=$rc wall $(( $(date +%s) - S0 )) s"; tail -5 gpurun_out/r3_bench_n2.err | cut -c1-300
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r3_bench_n2.json') if l.startswith('{')][-1])
print('n_gpus', d['n_gpus'], 'value', d['value'])
for k in ('fp32','b64','cifar','train','train_weak','cifar_big'):
    v=d.get(k); print(k, json.dumps({kk: v.get(kk) for kk in ('value','ms_per_step','error','global_batch','batch_per_gpu')}) if v else None)
PY
