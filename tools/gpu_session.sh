set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r3_new.log
timeout -k 10 900 python -u -m pytest -x -q -m gpu --timeout 400 --timeout-method thread -s \
  "tests/test_hip_training.py::test_fused_step_matches_torch_optimizer_tight_bar_in_the_deterministic_configuration" \
  "tests/test_hip_training.py::test_one_rank_rccl_rehearsal_matches_single_process" \
  "tests/test_hip_fullsize.py::test_large_batch_gradients_equal_mean_of_b128_slices" > $L 2>&1; rc=$?
grep -E "^\[|passed|failed|Error|error" $L | grep -v Gloo | cut -c1-400 | tail -30
if grep -q "HSA_STATUS_ERROR\|Memory access fault" $L; then echo FAULT; exit 1; fi
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python bench.py > gpurun_out/r3_bench0.json 2> gpurun_out/r3_bench0.err || { tail -20 gpurun_out/r3_bench0.err; exit 1; }
cat gpurun_out/r3_bench0.json | cut -c1-3000
