set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -u -m pytest tests/test_hip_training.py tests/test_hip_fullsize.py -x -q -k "autograd or ddp or golden" --timeout 400 --timeout-method thread -m gpu > gpurun_out/r2_gpu_all.log 2>&1; rc1=$?
tail -30 gpurun_out/r2_gpu_all.log | cut -c1-400
exit $rc1
