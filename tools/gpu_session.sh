set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_hip_training.py -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/ag_train.log 2>&1; rc=$?; tail -5 gpurun_out/ag_train.log | cut -c1-300
if grep -q "HSA_STATUS_ERROR\|Memory access fault" gpurun_out/ag_train.log; then echo FAULT; exit 1; fi
[ $rc -eq 0 ] || exit $rc
export GPT_DP_REHEARSE=1
for v in 0 1 0 1; do
echo "== GPT_DP_GATHER_OVERLAP=$v"
GPT_DP_GATHER_OVERLAP=$v timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$v bench.py --gpus 1 --mode train --batch 128 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('final_loss'))" || exit 1
done
