set -o pipefail
timeout -k 10 600 python -u -m pytest -x -q -m gpu --timeout 300 --timeout-method thread tests/test_hip_parity.py tests/test_hip_chains.py -k "sampl or loop or ddim or chain or seeded" 2>&1 | grep -E "^E  |passed|failed|Error" | head -20 || exit 1
for i in 1 2; do timeout -k 10 200 python bench.py --batch 64 --steps 400 --warmup 40 --no-train --no-extra --no-cpu-baseline --no-roofline --repeats 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('b64', d['ms_per_step'], d['spread']['ms_per_step_min'])" || exit 1; done
