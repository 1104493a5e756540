set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1200 python -u -m pytest tests/test_hip_training.py tests/test_hip_fullsize.py tests/test_hip_parity.py -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/cs.log 2>&1; rc=$?; tail -3 gpurun_out/cs.log | cut -c1-200
if grep -q "HSA_STATUS_ERROR\|Memory access fault" gpurun_out/cs.log; then echo FAULT; exit 1; fi
[ $rc -eq 0 ] || exit $rc
for v in 0 1 0 1; do
for b in 128 1024; do
echo "== GPT_DGELU_COLSUM=$v B=$b"
GPT_DGELU_COLSUM=$v timeout -k 10 300 python bench.py --mode train --batch $b --steps 20 --warmup 4 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('final_loss'))" || exit 1
done; done
