set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -u -m pytest tests -x -q --timeout 400 --timeout-method thread -m gpu > gpurun_out/r2_gpu_all.log 2>&1; rc1=$?
tail -3 gpurun_out/r2_gpu_all.log | cut -c1-300
[ $rc1 -eq 0 ] || exit $rc1
timeout -k 10 100 python -c "import __graft_entry__ as g; g.smoke()" || exit 1
timeout -k 10 600 python bench.py > gpurun_out/r2_bench1.json 2> gpurun_out/r2_bench1.err; rc2=$?
tail -3 gpurun_out/r2_bench1.err; echo "rc bench=$rc2"
timeout -k 10 300 python bench.py --batch 64 --no-train --no-extra --no-cpu-baseline > gpurun_out/r2_bench_b64.json 2>/dev/null
