set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1200 python -u -m pytest tests/test_hip_training.py tests/test_hip_fullsize.py tests/test_hip_parity.py -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/sg.log 2>&1; rc=$?; tail -3 gpurun_out/sg.log | cut -c1-200
if grep -q "HSA_STATUS_ERROR\|Memory access fault" gpurun_out/sg.log; then echo FAULT; exit 1; fi
[ $rc -eq 0 ] || exit $rc
for b in 128 1024; do
timeout -k 10 300 python bench.py --mode train --batch $b --steps 20 --warmup 4 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('final_loss'))" || exit 1
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/sgprof && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sgprof -- python bench.py --mode train --batch 1024 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/sgprof.log 2>&1 || exit 1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/sgprof/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'small_gemm' in r['Name'] or 'encode_kernel' in r['Name'] or 'decode_kernel' in r['Name']:
        print('  %-70s calls=%s avg_us=%.1f'%(r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
rm -rf gpurun_out/sgprof
