# dispatcher check at the configs[4] shapes (d = 3072, B = 68: M = 1768): default plan against forced tile choices
set -e
cd $GRAFT_REPO_ROOT
SH="qkv:1768:9216:3072:0 proj:1768:3072:3072:2 fc1:1768:12288:3072:1 fc2:1768:3072:12288:2 dfc2:1768:12288:3072:0 dfc1:1768:3072:12288:0 dqkv:1768:3072:9216:0"
for cfg in "X=0" "GPT_GEMM_TILE=3" "GPT_GEMM_TILE=2" "GPT_GEMM_TILE=1" "GPT_GEMM_224=0" "GPT_GEMM_192=0" "GPT_GEMM_192=2" "GPT_GEMM_SPLITK=0" "GPT_GEMM_TAIL128=0"; do
  echo "== $cfg"
  env $cfg timeout -k 10 120 python tools/gemm_bench.py --rounds 12 --shapes $SH
done
