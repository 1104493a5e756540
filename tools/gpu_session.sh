set -o pipefail
mkdir -p gpurun_out
python -c "import torch; print(torch.cuda.Stream.priority_range())"
for rep in 1 2; do
for h in 0 1; do
  for b in 128 1024; do
    GPT_TRAIN_HIPRIO=$h timeout -k 10 300 python bench.py --mode train --batch $b --steps 40 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/ovl_b.json 2> gpurun_out/ovl_b.err || { tail -5 gpurun_out/ovl_b.err; exit 1; }
    python -c "
import json
d=json.loads(open('gpurun_out/ovl_b.json').read().strip().splitlines()[-1])
print('hiprio',$h,'batch',$b,'ms',round(d['ms_per_step'],3),'loss',d.get('final_loss'))"
  done
done
done
