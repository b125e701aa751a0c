#!/bin/bash
mkdir -p gpurun_out/r4b4
export TMPDIR=/tmp
export SPK_TUNE_CACHE=$PWD/gpurun_out/r4b4/tune.txt
python3 bench.py --network efficientnet_b4 --batch 128 --mode train --no-cpu-baseline --steps 3 --warmup 2 > /dev/null 2>&1
rm -rf gpurun_out/r4b4/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4b4/prof -- python3 bench.py --network efficientnet_b4 --batch 128 --mode train --no-cpu-baseline --no-kernel-profile --steps 10 --warmup 3 > gpurun_out/r4b4/under.json 2> gpurun_out/r4b4/rocprof.err
f=$(find gpurun_out/r4b4/prof -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/r4b4/kernel_stats.csv
t=$(find gpurun_out/r4b4/prof -name "*kernel_trace.csv" | head -1)
python3 tools/step_timeline.py "$t" 40 > gpurun_out/r4b4/timeline.txt
rm -rf gpurun_out/r4b4/prof
head -60 gpurun_out/r4b4/timeline.txt
