# A/B of environment switches on a training configuration, one box: bash tools/ab_train_env.sh "<bench flags>" "VAR=v" "VAR=v" ...
set -e
cd $GRAFT_REPO_ROOT
flags=$1; shift
for rep in 1 2; do
for cfg in "$@"; do
  echo -n "$cfg rep $rep: "
  env $cfg timeout -k 10 280 python bench.py --mode train $flags --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])"
done; done
