set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_hip_parity.py tests/test_hip_fullsize.py -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/tok_parity.log 2>&1; rc=$?; tail -3 gpurun_out/tok_parity.log | cut -c1-200
if grep -q "HSA_STATUS_ERROR\|Memory access fault" gpurun_out/tok_parity.log; then echo FAULT; exit 1; fi
[ $rc -eq 0 ] || exit $rc
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1; do
for b in 64 320; do
rm -rf gpurun_out/tokprof && GPT_TOKENS_KC128=$v timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tokprof -- python bench.py --batch $b --steps 50 --warmup 5 --no-cpu-baseline --no-train --no-roofline --no-extra --repeats 1 > gpurun_out/tokprof.log 2>&1 || exit 1
echo "== KC128=$v B=$b"; python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/tokprof/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'encode_kernel' in r['Name'] or 'decode_kernel' in r['Name']:
        print('  %-60s calls=%s avg_us=%.1f'%(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
done; done
rm -rf gpurun_out/tokprof
