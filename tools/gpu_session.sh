set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_hip_chains.py -x -q --timeout 300 --timeout-method thread -m gpu > gpurun_out/r2_chains.log 2>&1; rc1=$?
tail -15 gpurun_out/r2_chains.log | cut -c1-300
exit $rc1
