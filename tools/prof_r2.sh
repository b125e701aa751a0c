#!/bin/bash
# Steady-state rocprofv3 kernel statistics of bench.py (infer and train separately) with a warm tuner cache, so that
# the CSV holds no autotuning launches.  Usage (on the GPU box, from the repo root): bash tools/prof_r2.sh <tag>
set -e
TAG=${1:-r02}
export TMPDIR=/tmp
CACHE=$PWD/gpurun_out/tune_${TAG}.txt
export SPK_TUNE_CACHE=$CACHE
# 1. warm the cache (tunes every problem of both modes once)
python3 bench.py --mode both --no-cpu-baseline --steps 3 --warmup 2 > gpurun_out/${TAG}_warm.json 2> gpurun_out/${TAG}_warm.err
for MODE in infer train; do
  rm -rf gpurun_out/prof_${TAG}_${MODE}
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_${MODE} -- python3 bench.py --mode $MODE --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/${TAG}_${MODE}_under_rocprof.json 2> gpurun_out/${TAG}_${MODE}_rocprof.err
  f=$(find gpurun_out/prof_${TAG}_${MODE} -name "*kernel_stats.csv" | head -1)
  cp "$f" gpurun_out/${TAG}_${MODE}_kernel_stats.csv
done
ls -la gpurun_out/${TAG}_*kernel_stats.csv
