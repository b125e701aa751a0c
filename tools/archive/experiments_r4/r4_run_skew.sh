#!/bin/bash
mkdir -p gpurun_out/r4skew
export SPK_TUNE_CACHE=$PWD/gpurun_out/r4skew/tune.txt
python3 bench.py --mode infer --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2>&1
for k in 0 6 12 16 20 26 32 40 100; do
  SPK_EVAL_SKEW=$k timeout -k 10 200 python bench.py --mode infer --no-cpu-baseline --no-kernel-profile --steps 40 --warmup 10 > gpurun_out/r4skew/bench_$k.json 2>gpurun_out/r4skew/bench_$k.err
  python -c "
import json; d=json.load(open('gpurun_out/r4skew/bench_$k.json')); print('skew $k', d['value'], d['ms_per_step'], d['step_ms'])"
done
