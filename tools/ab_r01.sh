run() { echo "== $*"; env "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('value'), d.get('final_loss'))"; }
run A=1 timeout -k 10 120 python bench.py --mode train --batch 128 --steps 20 --warmup 5 --no-cpu-baseline
run A=1 timeout -k 10 200 python bench.py --mode train --batch 1024 --steps 10 --warmup 3 --no-cpu-baseline
