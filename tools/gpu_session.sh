set -o pipefail
mkdir -p gpurun_out
SH="--rounds 40 --shapes fc1:8320:6144:1536:1 fc1d:8320:6144:1536:0"
for rep in 1 2; do
for k in 0 4 6 8; do
  echo "== tail128k=$k"; GPT_GEMM_TAIL128K=$k timeout -k 10 120 python tools/gemm_bench.py $SH 2>/dev/null || exit 1
done
done
