set -o pipefail
mkdir -p gpurun_out
S0=$(date +%s); timeout -k 10 500 python bench.py > gpurun_out/r3_bench3.json 2> gpurun_out/r3_bench3.err || { tail -20 gpurun_out/r3_bench3.err; exit 1; }
echo "bench wall: $(( $(date +%s) - S0 )) s"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3_bench3.json'))
print('headline', d['value'], d['ms_per_step'], d['roofline']['frac'])
print(json.dumps(d['hbm_kernels'], indent=1))
print('b64', d['b64']['ms_per_step'], 'train_weak', d['train_weak']['ms_per_step'], 'train', d['train']['ms_per_step'])
PY
