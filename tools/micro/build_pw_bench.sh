#!/bin/bash
# Builds tools/micro/pw_bench for gfx950 (cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
C=../../syke-pic_amd/csrc
mkdir -p build
for f in conv_pw conv_c3 conv_igemm conv_stem pointwise; do
  if [ ! -f build/$f.o ] || [ $C/$f.hip -nt build/$f.o ] || [ $C/spk_common.h -nt build/$f.o ]; then
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -I$C -c $C/$f.hip -o build/$f.o &
  fi
done
wait
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -I$C -c pw_bench.hip -o build/pw_bench.o
hipcc --offload-arch=gfx950 build/pw_bench.o build/conv_pw.o build/conv_c3.o build/conv_igemm.o build/conv_stem.o build/pointwise.o -o pw_bench
echo built pw_bench
