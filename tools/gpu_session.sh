set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest -x -q -m gpu --timeout 300 --timeout-method thread tests/test_hip_training.py tests/test_hip_fullsize.py -k "not two_rank and not more_ranks and not ddp and not rccl" 2>&1 | grep -E "^E  |passed|failed" | head || exit 1
for v in 0 1 0 1; do
  echo "GPT_LNB_PARTIALS=$v"; GPT_LNB_PARTIALS=$v timeout -k 10 200 python bench.py --mode train --batch 128 --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" || exit 1
done
