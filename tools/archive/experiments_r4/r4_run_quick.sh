#!/bin/bash
mkdir -p gpurun_out/r4q
timeout -k 10 900 python -m pytest tests/test_gpu_infer.py tests/test_gpu_calibrated.py tests/test_gpu_preprocess.py tests/test_gpu_workflows.py -m gpu -q -x > gpurun_out/r4q/test.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4q/test.txt | tail -5
export SPK_TUNE_CACHE=$PWD/gpurun_out/r4q/tune.txt
timeout -k 10 300 python bench.py --mode infer --no-cpu-baseline --layers-out gpurun_out/r4q/layers.json > gpurun_out/r4q/bench.json 2>gpurun_out/r4q/bench.err
python -c "
import json; d=json.load(open('gpurun_out/r4q/bench.json')); print('resnet50', d['value'], d['ms_per_step'], d['step_ms'], d['roofline']['frac'])
for l in json.load(open('gpurun_out/r4q/layers.json')):
    if l['layer'] in ('input.to_nhwc4','avgpool@base.8','base.0+maxpool'): print(l)"
