set -e
run() { echo "== $*"; env "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('value'))"; }
B="python bench.py --steps 150 --warmup 20 --no-cpu-baseline --no-train --no-roofline"
run A=1 timeout -k 10 120 $B --batch 64
run A=1 timeout -k 10 120 $B
