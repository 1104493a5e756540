set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1500 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/r2_gpu_all.log 2>&1; rc=$?; tail -3 gpurun_out/r2_gpu_all.log | cut -c1-300
if grep -q "HSA_STATUS_ERROR\|Memory access fault" gpurun_out/r2_gpu_all.log; then echo FAULT; exit 1; fi
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1 || { tail -5 gpurun_out/smoke.log; exit 1; }
tail -1 gpurun_out/smoke.log
