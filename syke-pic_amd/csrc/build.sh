#!/bin/bash
# Builds libsykepic_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
# The host-side sanitizer build lives in build_asan.sh (CPU only; that script does not travel to the GPU box).
set -e
cd "$(dirname "$0")"
OUT=../sykepic_hip/libsykepic_hip.so
SRCS="model.hip train.hip ops_abi.hip conv_igemm.hip conv_pw.hip conv_pwr.hip conv_c3.hip conv_bneck.hip conv_stem.hip conv_wgrad.hip pointwise.hip effnet.hip pw_fp8.hip dwconv_lds.hip preprocess.hip augment.hip head.hip train_kernels.hip train_effnet.hip zero_sum.hip"
mkdir -p build
pids=()
for f in $SRCS; do
  o=build/${f%.hip}.o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ spk_common.h -nt "$o" ] || [ model.h -nt "$o" ] || [ ../../include/sykepic_hip.h -nt "$o" ] || [ resize_u8.h -nt "$o" ] || [ ordered_reduce.h -nt "$o" ] || [ train_effnet.h -nt "$o" ] || [ dw_util.h -nt "$o" ]; then
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -c "$f" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait "$p"; done
objs=""
for f in $SRCS; do objs="$objs build/${f%.hip}.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o "$OUT"
echo "built $OUT"
