set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_hip_parity.py -x -q --timeout 200 --timeout-method thread -m gpu -k "gemm" > gpurun_out/r2_gemm_tests.log 2>&1; rc1=$?
tail -2 gpurun_out/r2_gemm_tests.log
[ $rc1 -eq 0 ] || exit $rc1
for rep in 1 2; do
for lib in libgpt_hip_nobal.so libgpt_hip.so; do
  echo "== $lib rep $rep"
  GPT_HIP_LIB=$PWD/g.pt_amd/lib/$lib timeout -k 10 120 python tools/gemm_bench.py --rounds 30 2>/dev/null || exit 1
done; done
