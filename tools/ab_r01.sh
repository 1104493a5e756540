set -e
run() { echo "== $*"; env "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('value'))"; }
B="python bench.py --steps 150 --warmup 20 --no-cpu-baseline --no-train --no-roofline"
OLD=GPT_HIP_LIB=$PWD/g.pt_amd/lib/libgpt_hip_old.so
run $OLD timeout -k 10 120 $B
run A=1 timeout -k 10 120 $B
run $OLD timeout -k 10 120 $B
run A=1 timeout -k 10 120 $B
