G="python tools/gemm_bench.py --shapes qkv:8320:4608:1536 fc2:8320:1536:6144:2 sq8192:8192:8192:8192"
for v in "GPT_GEMM_P=0" "GPT_GEMM_P=1" "GPT_GEMM_P=1 GPT_P_DBG=1" "GPT_GEMM_P=1 GPT_P_DBG=2"; do
  echo "== $v"; env $v GPT_GEMM_224=0 timeout -k 10 100 $G 2>&1 | grep -v amdgpu.ids
done
