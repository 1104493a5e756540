set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_hip_parity.py -x -q -m gpu -k "gemm" --timeout 120 --timeout-method thread > gpurun_out/g128_edges.log 2>&1; rc=$?; tail -3 gpurun_out/g128_edges.log | cut -c1-200
if grep -q "HSA_STATUS_ERROR\|Memory access fault" gpurun_out/g128_edges.log; then echo FAULT; exit 1; fi
exit $rc
