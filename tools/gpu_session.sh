set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1500 python -u -m pytest tests -x -q -m gpu --timeout 400 --timeout-method thread > gpurun_out/r3_gpu_all.log 2>&1; rc=$?; tail -3 gpurun_out/r3_gpu_all.log | cut -c1-300
if grep -q "HSA_STATUS_ERROR\|Memory access fault" gpurun_out/r3_gpu_all.log; then echo FAULT; exit 1; fi
[ $rc -eq 0 ] || { grep -E "^E |Error" gpurun_out/r3_gpu_all.log | head -20; exit $rc; }
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1 || { tail -5 gpurun_out/smoke.log; exit 1; }
tail -1 gpurun_out/smoke.log
S0=$(date +%s); timeout -k 10 500 python bench.py > gpurun_out/r3_bench4.json 2> gpurun_out/r3_bench4.err || { tail -20 gpurun_out/r3_bench4.err; exit 1; }
echo "bench wall: $(( $(date +%s) - S0 )) s"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3_bench4.json'))
print('headline', d['value'], d['ms_per_step'], d['roofline']['frac'])
for k in ('fp32','b64','cifar','train','train_weak','cifar_big'):
    v=d[k]; print(k, v.get('value'), v.get('ms_per_step'), (v.get('roofline') or {}).get('frac'), v.get('error'))
print({k:(v['cold']['achieved_GBps'], v['warm']['achieved_GBps']) for k,v in d['hbm_kernels'].items()})
print(d['cpu_baseline']['value'], d['cpu_baseline_train']['value'])
PY
