#!/bin/bash
mkdir -p gpurun_out/r4dy
timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_effnet_train.py -m gpu -q -x > gpurun_out/r4dy/test.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4dy/test.txt | tail -4
export SPK_TUNE_CACHE=$PWD/gpurun_out/r4dy/tune.txt
timeout -k 10 300 python bench.py --mode train --no-cpu-baseline --steps 5 --warmup 3 > /dev/null 2>&1
for v in 1 0 1 0; do
  SPK_DY_PER_LAYER=$v timeout -k 10 300 python bench.py --mode train --no-cpu-baseline --no-kernel-profile --steps 30 --warmup 8 > gpurun_out/r4dy/r50_$v.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r4dy/r50_$v.json')); print('resnet50 train per_layer=$v', d['value'], d['ms_per_step'])"
done
for v in 1 0; do
  SPK_DY_PER_LAYER=$v timeout -k 10 300 python bench.py --network efficientnet_b4 --batch 128 --mode train --no-cpu-baseline --no-kernel-profile --steps 20 --warmup 6 > gpurun_out/r4dy/b4_$v.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r4dy/b4_$v.json')); print('b4 train per_layer=$v', d['value'], d['ms_per_step'])"
done
