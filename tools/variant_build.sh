#!/bin/bash
# Builds a variant of the library for an A/B measurement: tools/variant_build.sh NAME "-DFOO=1 ..." file.hip [file.hip ...]
# compiles the named sources with the extra flags and links them with the product objects of everything else into
# g.pt_amd/lib/libgpt_hip_var_NAME.so (load it with GPT_HIP_LIB=...).  Variant libraries are not part of the product.
set -e
name=$1; flags=$2; shift 2
cd "$(dirname "$0")/../g.pt_amd/csrc"
make >/dev/null
mkdir -p ../lib/obj_var_$name
skip=""
for f in "$@"; do
  b=$(basename $f); b=${b%.*}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $flags -c $f -o ../lib/obj_var_$name/$b.o
  skip="$skip -e /$b.o"
done
objs=$(ls ../lib/obj/*.o | grep -v $skip)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../lib/libgpt_hip_var_$name.so $objs ../lib/obj_var_$name/*.o -ldl
echo built libgpt_hip_var_$name.so
