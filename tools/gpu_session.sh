set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest -x -q -m gpu --timeout 200 --timeout-method thread tests/test_hip_fullsize.py::test_ema_updated_through_dot_data_is_not_stale tests/test_hip_parity.py -k "ema_updated or forward or sampl" 2>&1 | grep -E "^E  |passed|failed" | head; 
for v in 64 320; do timeout -k 10 200 python tools/compat_step_bench.py $v 2>&1 | grep "ms per step" || exit 1; done
