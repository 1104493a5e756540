set -o pipefail
mkdir -p gpurun_out
for rep in 1 2; do
for r in 2 3 4; do
  GPT_LN_RPW=$r timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-train --no-extra --no-roofline --repeats 3 > gpurun_out/ln_b.json 2> gpurun_out/ln_b.err || { tail -5 gpurun_out/ln_b.err; exit 1; }
  python -c "
import json
d=json.loads(open('gpurun_out/ln_b.json').read().strip().splitlines()[-1])
print('rpw',$r,'ms',round(d['spread']['ms_per_step_median'],4), round(d['spread']['ms_per_step_min'],4))"
done
done
