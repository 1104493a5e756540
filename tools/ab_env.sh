# A/B of an environment switch on one box: bash tools/ab_env.sh VAR "v1 v2 ..." BATCH [extra bench flags]; two interleaved repetitions
set -e
cd $GRAFT_REPO_ROOT
var=$1; vals=$2; B=$3; shift 3
F="--no-cpu-baseline --no-train --no-roofline --no-extra --repeats 5"
for rep in 1 2; do
for v in $vals; do
  echo -n "$var=$v B=$B rep $rep: "
  env $var=$v timeout -k 10 150 python bench.py --batch $B $F "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['spread']['ms_per_step_median'])"
done; done
