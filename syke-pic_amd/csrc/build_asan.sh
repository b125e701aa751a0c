#!/bin/bash
# Host-side sanitizer build (CPU ONLY): every translation unit again with AddressSanitizer + UBSan on the HOST pass
# (GPU sanitizers are not available on this pool), linked with asan_driver.hip into build/asan/asan_driver, which
# tests/test_abi.py runs on the CPU.  Listed in .gpurunignore: neither this script nor its output goes to a GPU box.
set -e
cd "$(dirname "$0")"
mkdir -p build/asan
FL="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value -Xarch_host -fsanitize=address,undefined -Xarch_host -fno-omit-frame-pointer -Xarch_host -fno-sanitize-recover=undefined"
pids=()
for f in model.hip train.hip ops_abi.hip conv_igemm.hip conv_pw.hip conv_pwr.hip conv_c3.hip conv_bneck.hip conv_stem.hip conv_wgrad.hip pointwise.hip effnet.hip pw_fp8.hip dwconv_lds.hip preprocess.hip augment.hip head.hip train_kernels.hip train_effnet.hip zero_sum.hip asan_driver.hip; do
  o=build/asan/${f%.hip}.o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ spk_common.h -nt "$o" ] || [ model.h -nt "$o" ] || [ ../../include/sykepic_hip.h -nt "$o" ]; then
    hipcc $FL -c "$f" -o "$o" 2> "$o.log" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait "$p"; done
hipcc --offload-arch=gfx950 -fsanitize=address,undefined build/asan/*.o -o build/asan/asan_driver
echo "built build/asan/asan_driver"
