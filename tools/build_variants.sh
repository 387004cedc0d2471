#!/bin/bash
# tools/build_variants.sh name1 "-DFOO=1 -DBAR=2" name2 "..."  ->  variants/libumx_hip_<name>.so (A/B runs: UMX_HIP_LIB=...)
set -e
cd "$(dirname "$0")/../umx.cpp_amd"
mkdir -p ../variants
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function $flags -shared -o ../variants/libumx_hip_$name.so csrc/engine.hip &
done
wait
ls -la ../variants
