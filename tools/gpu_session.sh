set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -u -m pytest tests -x -q --timeout 400 --timeout-method thread -m gpu > gpurun_out/r2_gpu_all.log 2>&1; rc1=$?
tail -4 gpurun_out/r2_gpu_all.log
[ $rc1 -eq 0 ] || exit $rc1
timeout -k 10 600 python bench.py > gpurun_out/r2_bench0.json 2> gpurun_out/r2_bench0.err; rc2=$?
tail -c 3000 gpurun_out/r2_bench0.json; tail -5 gpurun_out/r2_bench0.err
echo "rc tests=$rc1 bench=$rc2"
