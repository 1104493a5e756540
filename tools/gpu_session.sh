set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_hip_training.py tests/test_hip_fullsize.py -x -q --timeout 300 --timeout-method thread -m gpu > gpurun_out/r2_train_tests.log 2>&1; rc1=$?
tail -4 gpurun_out/r2_train_tests.log | cut -c1-300
[ $rc1 -eq 0 ] || exit $rc1
for f in 0 1 0 1; do
  echo "== GPT_WGRAD_OVERWRITE=$f"
  GPT_WGRAD_OVERWRITE=$f timeout -k 10 200 python bench.py --mode train --batch 128 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['final_loss'])"
done
GPT_WGRAD_OVERWRITE=0 timeout -k 10 200 python bench.py --mode train --batch 1024 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b1024 off', d['ms_per_step'])"
GPT_WGRAD_OVERWRITE=1 timeout -k 10 200 python bench.py --mode train --batch 1024 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b1024 on', d['ms_per_step'])"
