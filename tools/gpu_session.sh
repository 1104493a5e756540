set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest -x -q -m gpu --timeout 200 --timeout-method thread tests/test_hip_parity.py::test_gemm_epilogues 2>&1 | tail -2 || exit 1
for v in 0 1 0 1; do
  echo "GPT_GEMM_192=$v"; GPT_GEMM_192=$v timeout -k 10 120 python tools/gemm_bench.py --shapes fc1:1664:6144:1536:1 big:1768:12288:3072:1 2>&1 | grep -v amdgpu.ids | tail -4 || exit 1
done
for v in 0 1 0 1; do
  echo "bench b64 GPT_GEMM_192=$v"; GPT_GEMM_192=$v timeout -k 10 200 python bench.py --batch 64 --steps 400 --warmup 40 --no-train --no-extra --no-cpu-baseline --no-roofline --repeats 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['spread']['ms_per_step_min'])" || exit 1
done
