#!/bin/bash
mkdir -p gpurun_out/r4join
export SPK_TUNE_CACHE=$PWD/gpurun_out/r4join/tune.txt
for k in 0 54 44 25 0 54 44 25; do
  SPK_EVAL_JOIN=$k timeout -k 10 300 python bench.py --mode infer --no-cpu-baseline --no-kernel-profile --steps 60 --warmup 12 > gpurun_out/r4join/bench_$k.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r4join/bench_$k.json')); print('join $k', d['value'], d['ms_per_step'], d['step_ms'])"
done
