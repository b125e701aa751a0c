#!/bin/bash
# scratch: c3 on the 64/128-channel 3x3 layers - tuner log, per-layer table, parity
mkdir -p gpurun_out/r4c3
export SPK_TUNE_CACHE=$PWD/gpurun_out/r4c3/tune.txt
rm -f $SPK_TUNE_CACHE
SPK_TUNE_LOG=1 python3 bench.py --mode infer --no-cpu-baseline --steps 20 --warmup 5 --layers-out gpurun_out/r4c3/layers.json > gpurun_out/r4c3/bench.json 2> gpurun_out/r4c3/bench.err
grep "3x3" gpurun_out/r4c3/bench.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r4c3/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
for l in json.load(open('gpurun_out/r4c3/layers.json')):
    if 'conv2' in l['layer']: print(l['layer'], l['ms'], l['tflops'])
PY
python3 -m pytest tests/test_gpu_calibrated.py tests/test_gpu_c3.py tests/test_gpu_infer.py -m gpu -x -q 2>&1 | tail -5
