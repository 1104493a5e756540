set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_hip_parity.py -x -q -m gpu -k "layernorm" --timeout 120 --timeout-method thread > gpurun_out/ln_tests.log 2>&1; rc=$?; tail -3 gpurun_out/ln_tests.log | cut -c1-300
[ $rc -eq 0 ] || { tail -30 gpurun_out/ln_tests.log | cut -c1-200; exit $rc; }
