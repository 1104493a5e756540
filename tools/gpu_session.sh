set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_hip_parity.py tests/test_hip_fullsize.py tests/test_hip_chains.py -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/p128.log 2>&1; rc=$?; tail -3 gpurun_out/p128.log | cut -c1-200
if grep -q "HSA_STATUS_ERROR\|Memory access fault" gpurun_out/p128.log; then echo FAULT; exit 1; fi
exit $rc
