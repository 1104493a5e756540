# sampling step time over batch sizes (mnist shapes, bf16): a look for cliffs in the GEMM dispatcher's plans
set -e
cd $GRAFT_REPO_ROOT
F="--no-cpu-baseline --no-train --no-roofline --no-extra --repeats 2 --steps 60 --warmup 10"
for B in ${BATCHES:-8 16 32 48 64 80 96 112 128 160 192 224 256 288 320 384 448 512 640 768 1024}; do
  timeout -k 10 120 python bench.py --batch $B $F ${EXTRA:-} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); B=$B
print('B=%5d  %8.3f ms/step  %7.2f us/row  %7.1f TF/s' % (B, d['ms_per_step'], d['ms_per_step']*1e3/(B*26), d['denoiser_tflops_per_gpu']))"
done
