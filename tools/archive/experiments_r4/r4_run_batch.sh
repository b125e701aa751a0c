#!/bin/bash
mkdir -p gpurun_out/r4bs
for b in 32 64 128 256 512; do
  timeout -k 10 300 python bench.py --mode infer --batch $b --no-cpu-baseline --no-kernel-profile --steps 40 --warmup 10 > gpurun_out/r4bs/bench_$b.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r4bs/bench_$b.json')); print('resnet50 batch $b', d['value'], d['ms_per_step'])"
done
