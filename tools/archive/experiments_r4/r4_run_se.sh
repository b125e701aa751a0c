#!/bin/bash
# scratch: squeeze-excitation scaling inside the project conv - parity, then B4 / B0 throughput with and without
mkdir -p gpurun_out/r4se
timeout -k 10 900 python -m pytest tests/test_gpu_effnet.py tests/test_gpu_fp8.py tests/test_gpu_c3.py -q -x -m gpu > gpurun_out/r4se/test.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4se/test.txt | tail -12
export SPK_TUNE_CACHE=$PWD/gpurun_out/r4se/tune.txt
for net in efficientnet_b4 efficientnet_b0; do for f in 1 0; do
  SPK_SE_FUSE=$f timeout -k 10 300 python bench.py --network $net --batch 128 --precision mixed --mode infer --no-cpu-baseline --layers-out gpurun_out/r4se/layers_${net}_f$f.json > gpurun_out/r4se/bench_${net}_f$f.json 2>gpurun_out/r4se/bench_${net}_f$f.err
  python -c "
import json; d=json.load(open('gpurun_out/r4se/bench_${net}_f$f.json')); print('$net SE_FUSE=$f', d['value'], d['ms_per_step'])"
done; done
