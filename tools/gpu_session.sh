set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_hip_training.py -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/bk.log 2>&1; rc=$?; tail -3 gpurun_out/bk.log | cut -c1-200
if grep -q "HSA_STATUS_ERROR\|Memory access fault" gpurun_out/bk.log; then echo FAULT; exit 1; fi
[ $rc -eq 0 ] || exit $rc
export GPT_DP_REHEARSE=1
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --mode train --batch 128 --steps 20 --warmup 4 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('final_loss'))"
