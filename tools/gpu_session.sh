set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_hip_training.py -x -q --timeout 200 --timeout-method thread -m gpu -k "native_rccl" > gpurun_out/r2_dp.log 2>&1; rc1=$?
tail -25 gpurun_out/r2_dp.log | cut -c1-300
exit $rc1
