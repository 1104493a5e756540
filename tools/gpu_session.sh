set -o pipefail
for v in 0 1 0 1; do echo "GPT_LN_FILL128=$v"; GPT_LN_FILL128=$v timeout -k 10 100 python tools/resadd_ln_bench.py 704 416 1024 2>&1 | grep "us" || exit 1; done
for v in 0 1 0 1; do
  echo "b64 GPT_LN_FILL128=$v"; GPT_LN_FILL128=$v timeout -k 10 200 python bench.py --batch 64 --steps 400 --warmup 40 --no-train --no-extra --no-cpu-baseline --no-roofline --repeats 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['spread']['ms_per_step_min'])" || exit 1
done
