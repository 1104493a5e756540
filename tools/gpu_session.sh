set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_hip_parity.py -x -q -m gpu --timeout 120 --timeout-method thread > gpurun_out/g128_parity.log 2>&1; rc=$?; tail -3 gpurun_out/g128_parity.log | cut -c1-200
if grep -q "HSA_STATUS_ERROR\|Memory access fault" gpurun_out/g128_parity.log; then echo FAULT; exit 1; fi
[ $rc -eq 0 ] || exit $rc
for v in 0 1 0 1; do
echo "== GPT_GEMM_128=$v"
GPT_GEMM_128=$v timeout -k 10 200 python tools/gemm_bench.py --rounds 30 --batch 64 2>&1 | grep -v amdgpu.ids | cut -c1-150 || exit 1
done
for v in 0 1; do
echo "== bench B=64 GPT_GEMM_128=$v"
GPT_GEMM_128=$v timeout -k 10 300 python bench.py --batch 64 --steps 100 --warmup 10 --no-cpu-baseline --no-extra --repeats 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'])" || exit 1
done
