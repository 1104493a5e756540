set -o pipefail
mkdir -p gpurun_out
for v in 0 1 0 1; do
echo "== GPT_GEMM_TAIL128=$v"
GPT_GEMM_TAIL128=$v timeout -k 10 200 python tools/gemm_bench.py --rounds 20 --shapes qkv320:8320:4608:1536 fc1_320:8320:6144:1536:1 qkv256:6656:4608:1536 fc1_256:6656:6144:1536:1 qkv512:13312:4608:1536 fc1_512:13312:6144:1536:1 cqkv:33280:6144:2048 cfc1:33280:8192:2048:1 cproj:33280:2048:2048:2 cfc2:33280:2048:8192:2 2>&1 | grep -v amdgpu.ids | cut -c1-110 || exit 1
done
