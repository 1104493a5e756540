set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r3_new.log
timeout -k 10 900 python -u -m pytest -x -q -m gpu --timeout 400 --timeout-method thread -s \
  "tests/test_hip_parity.py::test_actor_permutation_maps_bit_exact" "tests/test_hip_parity.py::test_cartpole_full_size_forward_vs_oracle" \
  "tests/test_hip_parity.py::test_forward_golden_fp32" "tests/test_hip_parity.py::test_forward_oracle_bf16" \
  "tests/test_hip_parity.py::test_permutation_maps_bit_exact" > $L 2>&1; rc=$?
grep -E "^\[|passed|failed|Error|error" $L | grep -v Gloo | cut -c1-400 | tail -30
if grep -q "HSA_STATUS_ERROR\|Memory access fault" $L; then echo FAULT; exit 1; fi
[ $rc -eq 0 ] || exit $rc
S0=$(date +%s); timeout -k 10 700 python bench.py > gpurun_out/r3_bench1.json 2> gpurun_out/r3_bench1.err || { tail -20 gpurun_out/r3_bench1.err; exit 1; }
echo "bench wall: $(( $(date +%s) - S0 )) s"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3_bench1.json'))
print('headline', d['value'], d['ms_per_step'], d['roofline']['frac'])
for k in ('fp32','b64','cifar','train','train_weak','cifar_big','cpu_baseline_train','cpu_baseline'):
    print(k, json.dumps(d.get(k))[:700])
PY
