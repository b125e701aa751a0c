#!/bin/bash
mkdir -p gpurun_out/r4str
export SPK_TUNE_CACHE=$PWD/gpurun_out/r4str/tune.txt
for st in 2 3 4 2 3 4; do
  SPK_EVAL_STREAMS=$st timeout -k 10 300 python bench.py --mode infer --no-cpu-baseline --no-kernel-profile --steps 40 --warmup 10 > gpurun_out/r4str/bench_r50_$st.json 2>gpurun_out/r4str/bench_r50_$st.err
  python -c "
import json; d=json.load(open('gpurun_out/r4str/bench_r50_$st.json')); print('resnet50 streams $st', d['value'], d['ms_per_step'], d['step_ms'])"
done
for st in 2 3 4; do
  SPK_EVAL_STREAMS=$st timeout -k 10 300 python bench.py --network efficientnet_b4 --batch 128 --precision mixed --mode infer --no-cpu-baseline --no-kernel-profile --steps 40 --warmup 10 > gpurun_out/r4str/bench_b4_$st.json 2>gpurun_out/r4str/bench_b4_$st.err
  python -c "
import json; d=json.load(open('gpurun_out/r4str/bench_b4_$st.json')); print('b4 streams $st', d['value'], d['ms_per_step'], d['step_ms'])"
done
