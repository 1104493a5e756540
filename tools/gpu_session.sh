set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_hip_training.py -x -q -m gpu -k "more_ranks" --timeout 300 --timeout-method thread > gpurun_out/dp4.log 2>&1; rc=$?; tail -4 gpurun_out/dp4.log | cut -c1-300
if grep -q "HSA_STATUS_ERROR\|Memory access fault" gpurun_out/dp4.log; then echo FAULT; exit 1; fi
exit $rc
