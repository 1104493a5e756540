set -o pipefail
mkdir -p gpurun_out
L=$PWD/g.pt_amd/lib
GPT_HIP_LIB=$L/libgpt_hip_agpr.so timeout -k 10 300 python -u -m pytest tests/test_hip_parity.py -x -q -m gpu -k "gemm" --timeout 120 --timeout-method thread > gpurun_out/agpr_parity.log 2>&1 || { tail -15 gpurun_out/agpr_parity.log; exit 1; }
tail -1 gpurun_out/agpr_parity.log
for rep in 1 2; do
echo "== base";      timeout -k 10 120 python tools/gemm_bench.py 2>/dev/null || exit 1
echo "== agpr";      GPT_HIP_LIB=$L/libgpt_hip_agpr.so timeout -k 10 120 python tools/gemm_bench.py 2>/dev/null || exit 1
done
echo "== base abl7"; GPT_HIP_LIB=$L/libgpt_hip_abl7.so timeout -k 10 120 python tools/gemm_bench.py --shapes sq8192:8192:8192:8192 2>/dev/null || exit 1
echo "== agpr abl7"; GPT_HIP_LIB=$L/libgpt_hip_agpr_abl7.so timeout -k 10 120 python tools/gemm_bench.py --shapes sq8192:8192:8192:8192 2>/dev/null || exit 1
