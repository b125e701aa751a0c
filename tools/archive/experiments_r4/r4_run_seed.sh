#!/bin/bash
mkdir -p gpurun_out/r4seed
SPK_TUNE_LOG=1 timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r4seed/bench.json 2> gpurun_out/r4seed/bench.err
grep -c "spk tune" gpurun_out/r4seed/bench.err
grep "spk tune" gpurun_out/r4seed/bench.err | head -5
ls -la ~/.cache/sykepic_hip/ | tail -3
python -c "
import json; d=json.load(open('gpurun_out/r4seed/bench.json')); print('default', d['value'], d['ms_per_step'], d['train']['value'], d['train']['ms_per_step'])"
SPK_TUNE_SEED=0 XDG_CACHE_HOME=/tmp/nocache timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r4seed/bench_noseed.json 2> gpurun_out/r4seed/bench_noseed.err
python -c "
import json; d=json.load(open('gpurun_out/r4seed/bench_noseed.json')); print('unseeded', d['value'], d['ms_per_step'], d['train']['value'], d['train']['ms_per_step'])"
