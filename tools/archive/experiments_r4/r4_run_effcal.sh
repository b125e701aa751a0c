#!/bin/bash
mkdir -p gpurun_out/r4ec
timeout -k 10 600 python tests/archive/diagnostics/effnet_calibrated.py > gpurun_out/r4ec/diag.txt 2> gpurun_out/r4ec/diag.err
cat gpurun_out/r4ec/diag.txt; tail -3 gpurun_out/r4ec/diag.err
export SPK_TUNE_CACHE=$PWD/gpurun_out/r4ec/tune.txt
for pr in mixed calibrated fast; do
  timeout -k 10 300 python bench.py --network efficientnet_b4 --batch 128 --precision $pr --mode infer --no-cpu-baseline --no-kernel-profile --steps 40 --warmup 10 > gpurun_out/r4ec/bench_b4_$pr.json 2>gpurun_out/r4ec/bench_b4_$pr.err
  python -c "
import json; d=json.load(open('gpurun_out/r4ec/bench_b4_$pr.json')); print('b4 $pr', d['value'], d['ms_per_step'])"
done
