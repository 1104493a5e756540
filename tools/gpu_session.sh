set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1500 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/r2_gpu_all.log 2>&1; rc=$?; tail -3 gpurun_out/r2_gpu_all.log | cut -c1-300
if grep -q "HSA_STATUS_ERROR\|Memory access fault" gpurun_out/r2_gpu_all.log; then echo FAULT; exit 1; fi
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1 || { tail -5 gpurun_out/smoke.log; exit 1; }
tail -1 gpurun_out/smoke.log
timeout -k 10 900 python bench.py > gpurun_out/r2_bench12.json 2> gpurun_out/r2_bench12.err || { tail -5 gpurun_out/r2_bench12.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2_bench12.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, 'frac', d['roofline']['frac'])
for k in ('fp32','cifar','train','train_weak'):
    o=d.get(k); 
    if o: print(k,{kk:(o[kk] if kk!='roofline' else o[kk]['frac']) for kk in o if kk in ('value','ms_per_step','roofline','error')})
PY
bash tools/profile_r02.sh > gpurun_out/profile_r02.log 2>&1 || { tail -5 gpurun_out/profile_r02.log; exit 1; }
grep "GPU busy" gpurun_out/profile_r02.log
timeout -k 10 300 python bench.py --batch 64 --steps 300 --warmup 30 --no-cpu-baseline --no-train --no-extra --repeats 3 > gpurun_out/r2_b64.json 2> gpurun_out/r2_b64.err || exit 1
python -c "
import json
d=json.loads(open('gpurun_out/r2_b64.json').read().strip().splitlines()[-1])
print('B=64', d['value'], d['ms_per_step'], d['roofline']['frac'])"
