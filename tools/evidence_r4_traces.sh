#!/bin/bash
# the four kernel traces of tools/evidence_r4.sh alone (warm tuner cache from that run, or re-tuned here)
set -e
TAG=r04
export TMPDIR=/tmp
mkdir -p gpurun_out
export SPK_TUNE_CACHE=$PWD/gpurun_out/tune_${TAG}.txt
[ -f gpurun_out/${TAG}_tune_cache.txt ] && cp gpurun_out/${TAG}_tune_cache.txt $SPK_TUNE_CACHE
python3 bench.py --mode both --no-cpu-baseline --steps 3 --warmup 2 > /dev/null 2>&1
trace() {
  local NAME=$1 MODE=$2; shift 2
  rm -rf gpurun_out/prof_${TAG}_${NAME}
  ( for kv in "$@"; do export "$kv"; done
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_${NAME} -- python3 bench.py --mode $MODE --no-cpu-baseline --no-kernel-profile --steps 20 --warmup 5 > gpurun_out/${TAG}_${NAME}_under_rocprof.json 2> gpurun_out/${TAG}_${NAME}_rocprof.err )
  local f=$(find gpurun_out/prof_${TAG}_${NAME} -name "*kernel_stats.csv" | head -1)
  cp "$f" gpurun_out/${TAG}_${NAME}_kernel_stats.csv
  local t=$(find gpurun_out/prof_${TAG}_${NAME} -name "*kernel_trace.csv" | head -1)
  python3 tools/step_timeline.py "$t" 30 > gpurun_out/${TAG}_timeline_${NAME}.txt
  rm -rf gpurun_out/prof_${TAG}_${NAME}
  echo "kernel trace $NAME done"
}
trace infer_2streams infer SPK_EVAL_STREAMS=2
trace infer_1stream infer SPK_EVAL_STREAMS=1
trace train_2streams train SPK_WGRAD_STREAM=1
trace train_1stream train SPK_WGRAD_STREAM=0
head -6 gpurun_out/${TAG}_timeline_infer_2streams.txt; head -6 gpurun_out/${TAG}_timeline_train_2streams.txt
