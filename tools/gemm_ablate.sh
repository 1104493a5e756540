#!/bin/bash
# Builds diagnostic variants of the 256x256 GEMM main loop (GPT_ABL bits: 1 no DMA, 2 no fragment reads, 4 no barriers; results are
# garbage, only the timing means something) next to the product library: g.pt_amd/lib/libgpt_hip_abl<N>.so
set -e
cd "$(dirname "$0")/../g.pt_amd/csrc"
make >/dev/null
FILE=${FILE:-gemm256}            # FILE=gemm128: the 128 x 256 kernel (GPT_ABL128 bits, + 8 = no MFMAs)
DEF=GPT_ABL; [ "$FILE" = gemm128 ] && DEF=GPT_ABL128
for abl in "$@"; do
  mkdir -p ../lib/obj_abl$abl
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-but-set-variable -Wno-unused-variable -D$DEF=$abl -c $FILE.hip -o ../lib/obj_abl$abl/$FILE.o
  objs=$(ls ../lib/obj/*.o | grep -v "/$FILE.o")
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../lib/libgpt_hip_abl$abl.so $objs ../lib/obj_abl$abl/$FILE.o -ldl
  echo built abl$abl
done
