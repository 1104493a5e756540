# One MI355X session: the whole GPU suite, smoke() and the default bench line (run through gpurun from the repo root).
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1500 python -u -m pytest tests -x -q -m gpu --timeout 400 --timeout-method thread > gpurun_out/r3_gpu_all.log 2>&1; rc=$?; tail -3 gpurun_out/r3_gpu_all.log | cut -c1-300
if grep -q "HSA_STATUS_ERROR\|Memory access fault" gpurun_out/r3_gpu_all.log; then echo FAULT; exit 1; fi
[ $rc -eq 0 ] || { grep -E "^E |Error" gpurun_out/r3_gpu_all.log | head -20; exit $rc; }
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1 || { tail -5 gpurun_out/smoke.log; exit 1; }
tail -1 gpurun_out/smoke.log
timeout -k 10 500 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err || { tail -20 gpurun_out/bench.err; exit 1; }
cut -c1-600 gpurun_out/bench.json
