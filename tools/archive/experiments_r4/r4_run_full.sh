#!/bin/bash
# scratch: whole GPU suite, then the EfficientNet training lines
mkdir -p gpurun_out/r4full
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r4full/test.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4full/test.txt | tail -8
export SPK_TUNE_CACHE=$PWD/gpurun_out/r4full/tune.txt
for net in efficientnet_b0 efficientnet_b4; do
  timeout -k 10 300 python bench.py --network $net --batch 128 --mode train --no-cpu-baseline > gpurun_out/r4full/bench_train_${net}.json 2>gpurun_out/r4full/bench_train_${net}.err
  python -c "
import json; d=json.load(open('gpurun_out/r4full/bench_train_${net}.json')); print('$net train', d['value'], d['ms_per_step'], d['roofline'].get('phases_ms'))"
done
