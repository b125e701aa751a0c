#!/bin/bash
mkdir -p gpurun_out/r4l
export SPK_TUNE_CACHE=$PWD/gpurun_out/r4l/tune.txt
for pr in mixed fp8; do
  timeout -k 10 300 python bench.py --network efficientnet_b4 --batch 128 --precision $pr --mode infer --no-cpu-baseline --layers-out gpurun_out/r4l/layers_$pr.json > gpurun_out/r4l/bench_$pr.json 2>gpurun_out/r4l/bench_$pr.err
  python -c "
import json; d=json.load(open('gpurun_out/r4l/bench_$pr.json')); print('b4 $pr', d['value'], d['ms_per_step'])"
done
