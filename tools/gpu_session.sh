set -o pipefail
for v in 1 2 1 2; do
  echo "GPT_GEMM_192=$v"; GPT_GEMM_192=$v timeout -k 10 120 python tools/gemm_bench.py --shapes fc1:3328:6144:1536:1 dg:3328:6144:1536:0 fc1b:3520:6144:1536:1 x:6656:6144:1536:1 2>&1 | grep -v amdgpu.ids | tail -5 || exit 1
done
