set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_hip_training.py -x -q -m gpu -k "overlapped_optimizer or store_mode or fused_step or golden" --timeout 120 --timeout-method thread > gpurun_out/ovl_test.log 2>&1 || { tail -30 gpurun_out/ovl_test.log; exit 1; }
tail -1 gpurun_out/ovl_test.log
for args in "--case cifar_big --batch 68 --augment" "--case mnist --batch 96 --dtype fp32" "--case mnist --batch 100"; do
  for o in 0 1; do
    GPT_OPT_OVERLAP=$o timeout -k 10 300 python bench.py --mode train $args --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/t.json 2> gpurun_out/t.err || { tail -5 gpurun_out/t.err; exit 1; }
    python -c "
import json
d=json.loads(open('gpurun_out/t.json').read().strip().splitlines()[-1])
print('$args', 'overlap', $o, 'ms', round(d['ms_per_step'],3), 'loss', d.get('final_loss'))"
  done
done
