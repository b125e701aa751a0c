#!/bin/bash
mkdir -p gpurun_out/r4e2
timeout -k 10 1000 python -m pytest tests/test_gpu_effnet.py tests/test_gpu_fp8.py tests/test_gpu_trained.py tests/test_gpu_workflows.py tests/test_gpu_effnet_train.py -m gpu -q > gpurun_out/r4e2/test.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4e2/test.txt | tail -15
export SPK_TUNE_CACHE=$PWD/gpurun_out/r4e2/tune.txt
for net in efficientnet_b4 efficientnet_b0; do for pr in mixed fp8; do
  timeout -k 10 300 python bench.py --network $net --batch 128 --precision $pr --mode infer --no-cpu-baseline > gpurun_out/r4e2/bench_${net}_$pr.json 2>gpurun_out/r4e2/bench_${net}_$pr.err
  python -c "
import json; d=json.load(open('gpurun_out/r4e2/bench_${net}_$pr.json')); print('$net $pr', d['value'], d['ms_per_step'])"
done; done
