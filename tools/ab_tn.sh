set -e
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in base tnold; do
  lib=$GRAFT_REPO_ROOT/g.pt_amd/lib/libgpt_hip_var_$v.so; [ "$v" = base ] && lib=$GRAFT_REPO_ROOT/g.pt_amd/lib/libgpt_hip.so
  echo "== $v rep $rep"
  GPT_HIP_LIB=$lib timeout -k 10 120 python tools/tn_bench.py 26624 | cut -c1-60
  GPT_HIP_LIB=$lib timeout -k 10 120 python tools/tn_bench.py 3328 | cut -c1-60
done; done
timeout -k 10 300 python -u -m pytest tests/test_hip_training.py -x -q -m gpu --timeout 300 --timeout-method thread -k "gemm_tn or wgrad_tn or pruned_last" 2>&1 | tail -2
