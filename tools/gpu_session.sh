set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_hip_parity.py tests/test_hip_fullsize.py -x -q -m gpu -k "resadd or forward or fullsize or full_size" --timeout 300 --timeout-method thread > gpurun_out/red_tests.log 2>&1; rc=$?; tail -2 gpurun_out/red_tests.log | cut -c1-300
[ $rc -eq 0 ] || { tail -30 gpurun_out/red_tests.log | cut -c1-200; exit $rc; }
for rep in 1 2; do
for v in 0 2 3 6; do
  GPT_LN_SPLITROW=$v timeout -k 10 300 python bench.py --batch 64 --steps 300 --warmup 30 --no-cpu-baseline --no-train --no-extra --no-roofline --repeats 3 > gpurun_out/red_b.json 2> gpurun_out/red_b.err || { tail -5 gpurun_out/red_b.err; exit 1; }
  python -c "
import json
d=json.loads(open('gpurun_out/red_b.json').read().strip().splitlines()[-1])
print('splitrow $v sample B=64 ms',round(d['spread']['ms_per_step_median'],4), round(d['spread']['ms_per_step_min'],4))"
  GPT_LN_SPLITROW=$v timeout -k 10 300 python bench.py --mode train --batch 128 --steps 40 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/red_t.json 2> gpurun_out/red_t.err || { tail -5 gpurun_out/red_t.err; exit 1; }
  python -c "
import json
d=json.loads(open('gpurun_out/red_t.json').read().strip().splitlines()[-1])
print('splitrow $v train B=128 ms',round(d['ms_per_step'],3))"
done
done
