#!/bin/bash
# The un-profiled bench lines of round 5 (run after tools/evidence_r5.sh has been copied into profiles/: the default line
# quotes the PMC traffic committed just before).  Default tuner cache of a fresh box = the shipped seed.
set -e
mkdir -p gpurun_out
python3 bench.py --layers-out gpurun_out/r05_infer_calibrated_layers_line.json > gpurun_out/r05_bench_line.json 2> gpurun_out/r05_lines.err
python3 bench.py --network resnet18 --batch 512 > gpurun_out/r05_bench_line_resnet18_b512.json 2>> gpurun_out/r05_lines.err
python3 bench.py --mode infer --network efficientnet_b4 --batch 128 > gpurun_out/r05_bench_efficientnet_b4_mixed_b128.json 2>> gpurun_out/r05_lines.err
python3 bench.py --mode infer --network efficientnet_b4 --batch 128 --precision fp8 --no-cpu-baseline > gpurun_out/r05_bench_efficientnet_b4_fp8_b128.json 2>> gpurun_out/r05_lines.err
python3 bench.py --mode train --network efficientnet_b4 --batch 128 --no-cpu-baseline > gpurun_out/r05_bench_efficientnet_b4_train_b128.json 2>> gpurun_out/r05_lines.err
for i in 1 2 3; do python3 bench.py --no-cpu-baseline --no-kernel-profile 2>> gpurun_out/r05_lines.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('repeat', d['ms_per_step'], d['value'], d['train']['ms_per_step'])"; done
python3 tools/bneck_bench.py > gpurun_out/r05_bneck_bench_final.txt 2>&1
for f in r05_bench_line r05_bench_line_resnet18_b512 r05_bench_efficientnet_b4_mixed_b128 r05_bench_efficientnet_b4_fp8_b128 r05_bench_efficientnet_b4_train_b128; do python3 -c "
import json; d=json.loads(open('gpurun_out/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('traffic'), (d.get('train') or {}).get('ms_per_step'))"; done
