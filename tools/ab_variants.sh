# A/B of variant libraries (tools/variant_build.sh) on one box: the GEMM micro-benchmark and the B = 320 sampling step, interleaved.
# usage: bash tools/ab_variants.sh NAME [NAME ...]   ("base" = the product library)
set -e
cd $GRAFT_REPO_ROOT
F="--no-cpu-baseline --no-train --no-roofline --no-extra --repeats 3"
for rep in 1 2; do
for v in "$@"; do
  lib=$GRAFT_REPO_ROOT/g.pt_amd/lib/libgpt_hip_var_$v.so; [ "$v" = base ] && lib=$GRAFT_REPO_ROOT/g.pt_amd/lib/libgpt_hip.so
  echo "== $v rep $rep"
  GPT_HIP_LIB=$lib timeout -k 10 120 python tools/gemm_bench.py --batch ${BATCH:-320} --rounds 20 | head -4
  GPT_HIP_LIB=$lib timeout -k 10 120 python bench.py --batch ${BATCH:-320} $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('step ms', d['ms_per_step'], d['spread']['ms_per_step_median'])"
done; done
