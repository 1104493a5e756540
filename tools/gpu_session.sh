set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_hip_fullsize.py -x -q -m gpu -k "deferred" --timeout 300 --timeout-method thread > gpurun_out/defer_t.log 2>&1; rc=$?; tail -15 gpurun_out/defer_t.log | cut -c1-220
if grep -q "HSA_STATUS_ERROR\|Memory access fault" gpurun_out/defer_t.log; then echo FAULT; exit 1; fi
exit $rc
