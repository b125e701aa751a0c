#!/bin/bash
mkdir -p gpurun_out/r4se
timeout -k 10 900 python -m pytest tests/test_gpu_effnet.py tests/test_gpu_fp8.py tests/test_gpu_infer.py -q -x -m gpu > gpurun_out/r4se/test3.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4se/test3.txt | tail -5
export SPK_TUNE_CACHE=$PWD/gpurun_out/r4se/tune3.txt
for net in efficientnet_b4; do for pr in mixed fp8; do for f in 1 0; do
  SPK_SE_SMALL=$f timeout -k 10 300 python bench.py --network $net --batch 128 --precision $pr --mode infer --no-cpu-baseline > gpurun_out/r4se/bench3_${net}_${pr}_small$f.json 2>gpurun_out/r4se/bench3.err
  python -c "
import json; d=json.load(open('gpurun_out/r4se/bench3_${net}_${pr}_small$f.json')); print('$net $pr SE_SMALL=$f', d['value'], d['ms_per_step'])"
done; done; done
timeout -k 10 300 python bench.py --mode infer --no-cpu-baseline > gpurun_out/r4se/bench3_r50.json 2>gpurun_out/r4se/bench3_r50.err
python -c "
import json; d=json.load(open('gpurun_out/r4se/bench3_r50.json')); print('resnet50', d['value'], d['ms_per_step'])"
