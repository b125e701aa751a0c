#!/bin/bash
# the bench lines of the final tree on one box: default line (quotes the committed PMC traffic), ResNet-18 b512, EfficientNets
mkdir -p gpurun_out/r4lines
export SPK_TUNE_CACHE=$PWD/gpurun_out/r4lines/tune.txt
python3 bench.py --mode both --no-cpu-baseline --steps 3 --warmup 2 > /dev/null 2>&1
python3 bench.py --layers-out gpurun_out/r4lines/r04_infer_calibrated_layers.json > gpurun_out/r4lines/r04_bench_line.json 2> gpurun_out/r4lines/bench.err
python3 -c "
import json; d=json.load(open('gpurun_out/r4lines/r04_bench_line.json')); print('default', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_frac'], d['roofline']['traffic'], d['train']['value'], d['train']['ms_per_step'])"
python3 bench.py --network resnet18 --batch 512 > gpurun_out/r4lines/r04_bench_line_resnet18_b512.json 2>> gpurun_out/r4lines/bench.err
python3 -c "
import json; d=json.load(open('gpurun_out/r4lines/r04_bench_line_resnet18_b512.json')); print('resnet18 b512', d['value'], d['ms_per_step'], d['roofline']['frac'], d['train']['value'])"
for net in efficientnet_b4 efficientnet_b0; do for pr in mixed fp8; do
  python3 bench.py --network $net --batch 128 --precision $pr --no-cpu-baseline > gpurun_out/r4lines/r04_bench_${net}_${pr}_b128.json 2>> gpurun_out/r4lines/bench.err
  python3 -c "
import json; d=json.load(open('gpurun_out/r4lines/r04_bench_${net}_${pr}_b128.json')); print('$net $pr', d['value'], d['ms_per_step'], d.get('train',{}).get('value'), d.get('train',{}).get('ms_per_step'))"
done; done
python3 bench.py --network efficientnet_b4 --batch 256 --precision mixed --mode infer --no-cpu-baseline > gpurun_out/r4lines/r04_bench_efficientnet_b4_mixed_b256.json 2>> gpurun_out/r4lines/bench.err
python3 -c "
import json; d=json.load(open('gpurun_out/r4lines/r04_bench_efficientnet_b4_mixed_b256.json')); print('b4 b256', d['value'], d['ms_per_step'])"
