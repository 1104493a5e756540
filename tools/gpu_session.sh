set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/ablate.txt
for v in "" _abl1 _abl2 _abl3 _abl4 _abl7 ""; do
  echo "== variant '$v'" >> gpurun_out/ablate.txt
  GPT_HIP_LIB=$PWD/g.pt_amd/lib/libgpt_hip$v.so timeout -k 10 200 python tools/gemm_bench.py --rounds 20 --shapes sq8192:8192:8192:8192 qkv:8320:4608:1536 fc2s:8320:1536:6144 >> gpurun_out/ablate.txt 2>&1 || exit 1
done
grep -v amdgpu.ids gpurun_out/ablate.txt | cut -c1-150
