set -o pipefail
for rep in 1 2; do for gq in 1 2; do for zg in 1 0; do GPT_OPT_GROUPS=$gq ZG=$zg timeout -k 10 120 python tools/opt_bench.py 2>/dev/null || exit 1; done; done; done
timeout -k 10 300 python -u -m pytest tests/test_hip_training.py -x -q --timeout 200 --timeout-method thread -m gpu -k "golden or fused or store_mode" 2>&1 | tail -2
