set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 560 python -u -m pytest tests/test_hip_parity.py -x -q -m gpu --timeout 300 --timeout-method thread -k "paired_input or forward or p_sample_loop or seeded_loop or full_size_sampling" > gpurun_out/ln_pair_tests.log 2>&1
tail -3 gpurun_out/ln_pair_tests.log
F="--no-cpu-baseline --no-train --no-roofline --no-extra --repeats 5"
for rep in 1 2; do
for B in 64 320; do
for flag in 1 0; do
  echo "B=$B FUSE0=$flag rep=$rep"
  GPT_LN_FUSE0=$flag timeout -k 10 120 python bench.py --batch $B $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d.get('ms_per_step_repeats'))"
done; done; done
