set -o pipefail
timeout -k 10 600 python -u -m pytest -x -q -m gpu --timeout 300 --timeout-method thread tests/test_hip_training.py::test_pruned_last_block_training_equals_all_rows 2>&1 | grep -E "^E  |passed|failed|Error" | head -20
