set -o pipefail
mkdir -p gpurun_out
for rep in 1 2; do
for c in 256 416 512; do
  for b in 128 1024; do
  GPT_LNB_BLOCKS=$c timeout -k 10 300 python bench.py --mode train --batch $b --steps 40 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/red_t.json 2> gpurun_out/red_t.err || { tail -5 gpurun_out/red_t.err; exit 1; }
  python -c "
import json
d=json.loads(open('gpurun_out/red_t.json').read().strip().splitlines()[-1])
print('lnb blocks $c train B=$b ms',round(d['ms_per_step'],3))"
  done
done
done
