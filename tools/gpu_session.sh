set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_hip_training.py tests/test_zz_no_alloc.py -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/train_tests.log 2>&1; rc=$?; tail -3 gpurun_out/train_tests.log | cut -c1-300
if grep -q "HSA_STATUS_ERROR\|Memory access fault" gpurun_out/train_tests.log; then echo FAULT; exit 1; fi
exit $rc
