set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest -x -q -m gpu --timeout 300 --timeout-method thread tests/test_hip_parity.py tests/test_hip_chains.py tests/test_hip_fullsize.py -k "not training and not gradients" 2>&1 | grep -E "^E  |passed|failed" | head || exit 1
for v in 0 1 0 1; do
  echo "GPT_PRUNE_LAST=$v"; GPT_PRUNE_LAST=$v timeout -k 10 200 python bench.py --steps 300 --warmup 30 --no-train --no-extra --no-cpu-baseline --no-roofline --repeats 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['spread']['ms_per_step_min'])" || exit 1
done
for v in 0 1; do
  echo "b64 GPT_PRUNE_LAST=$v"; GPT_PRUNE_LAST=$v timeout -k 10 200 python bench.py --batch 64 --steps 400 --warmup 40 --no-train --no-extra --no-cpu-baseline --no-roofline --repeats 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['spread']['ms_per_step_min'])" || exit 1
  echo "cifar GPT_PRUNE_LAST=$v"; GPT_PRUNE_LAST=$v timeout -k 10 200 python bench.py --case cifar --batch 1280 --steps 40 --warmup 5 --no-train --no-extra --no-cpu-baseline --no-roofline --repeats 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['spread']['ms_per_step_min'])" || exit 1
done
