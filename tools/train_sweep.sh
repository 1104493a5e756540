# training step time over per-GPU batch sizes (mnist shapes, bf16)
set -e
cd $GRAFT_REPO_ROOT
for B in ${BATCHES:-16 32 64 100 128 160 192 256 384 512 768 1024}; do
  timeout -k 10 200 python bench.py --mode train --batch $B --steps 12 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); B=$B
print('B=%5d  %8.3f ms/step  %7.2f us/row  %7.1f TF/s model' % (B, d['ms_per_step'], d['ms_per_step']*1e3/(B*26), d['model_tflops_per_gpu']))"
done
