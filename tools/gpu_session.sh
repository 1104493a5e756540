set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python -u -m pytest tests/test_bench_ranks.py -x -v -m gpu --timeout 300 --timeout-method thread > gpurun_out/bench_ranks.log 2>&1; rc=$?; tail -6 gpurun_out/bench_ranks.log | cut -c1-300
exit $rc
