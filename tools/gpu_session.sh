set -o pipefail
SH="qkv64:1664:4608:1536:0 longk:1664:4608:6144:0"
echo "abl 0 (product)"; timeout -k 10 100 python tools/gemm_bench.py --shapes $SH 2>&1 | grep -v amdgpu.ids | tail -2
for a in 1 2 3 4 7 8 9 10; do echo "abl $a"; GPT_HIP_LIB=$PWD/g.pt_amd/lib/libgpt_hip_abl$a.so timeout -k 10 100 python tools/gemm_bench.py --shapes $SH 2>&1 | grep -v amdgpu.ids | tail -2 || exit 1; done
