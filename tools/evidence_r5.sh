#!/bin/bash
# Round-5 evidence on one GPU box, in one call (run LAST, after the final source edit: the PMC traffic files are stamped
# with the source digest bench.py checks).  Usage (repo root): SPK_COMMIT=<sha> bash tools/evidence_r5.sh
#   1. warm the tuner cache; 2. the default bench line (inference `calibrated` + train + cpu_baseline, per-layer table);
#   3. rocprofv3 --kernel-trace --stats of the same commands, warm cache: the DEFAULT configuration (two eval streams,
#      weight gradients on their own stream - what the driver times: tools/step_timeline.py per-queue tables) AND the
#      one-stream runs (a kernel's duration is its own only when nothing overlaps it: the per-kernel stats);
#   4. PMC passes: FETCH_SIZE / WRITE_SIZE (infer and train), SQ MFMA-busy split.
set -e
TAG=r05
export TMPDIR=/tmp
mkdir -p gpurun_out
export SPK_TUNE_CACHE=$PWD/gpurun_out/tune_${TAG}.txt
rm -f $SPK_TUNE_CACHE
# start from the shipped tuner seed - what a fresh box (the driver's) starts from - and tune whatever it does not list
grep -v "^#" syke-pic_amd/sykepic_hip/tune_seed_gfx950.txt > $SPK_TUNE_CACHE
python3 bench.py --mode both --no-cpu-baseline --steps 3 --warmup 2 > gpurun_out/${TAG}_warm.json 2> gpurun_out/${TAG}_warm.err
cp $SPK_TUNE_CACHE gpurun_out/${TAG}_tune_cache.txt
python3 bench.py --layers-out gpurun_out/${TAG}_infer_calibrated_layers.json > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench_line.err
echo "bench line done"
trace() {   # name, mode, extra env assignments...
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
# counters: one stream each (a kernel's bytes are its own either way, but the dispatch order of one step is then the layer order)
export SPK_EVAL_STREAMS=1
export SPK_WGRAD_STREAM=0
for MODE in infer train; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_${TAG}_${MODE}_$C
    rocprofv3 --pmc $C --output-format csv -d gpurun_out/pmc_${TAG}_${MODE}_$C -- python3 bench.py --mode $MODE --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> gpurun_out/pmc_${TAG}_${MODE}_$C.err
    echo "pmc $MODE $C done"
  done
  F=$(find gpurun_out/pmc_${TAG}_${MODE}_FETCH_SIZE -name "*counter_collection.csv" | head -1)
  W=$(find gpurun_out/pmc_${TAG}_${MODE}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
  NAME=$([ $MODE = infer ] && echo infer_calibrated || echo train_bf16)
  python3 tools/pmc_traffic.py "$F" "$W" gpurun_out/${TAG}_pmc_traffic_${NAME}.json $MODE > gpurun_out/${TAG}_pmc_traffic_${MODE}.log
  rm -rf gpurun_out/pmc_${TAG}_${MODE}_FETCH_SIZE gpurun_out/pmc_${TAG}_${MODE}_WRITE_SIZE
done
rm -rf gpurun_out/pmc_${TAG}_sq
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc_${TAG}_sq -- python3 bench.py --mode infer --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> gpurun_out/pmc_${TAG}_sq.err
S=$(find gpurun_out/pmc_${TAG}_sq -name "*counter_collection.csv" | head -1)
python3 tools/pmc_mfma.py "$S" gpurun_out/${TAG}_pmc_mfma_util_infer_calibrated.json > gpurun_out/${TAG}_pmc_mfma.log
rm -rf gpurun_out/pmc_${TAG}_sq
tail -n 4 gpurun_out/${TAG}_pmc_traffic_infer.log; tail -n 4 gpurun_out/${TAG}_pmc_mfma.log
# per-kernel SQ counters (three --pmc passes, one stream): wave-cycle split, issue mix, LDS bank conflicts
unset SPK_EVAL_STREAMS SPK_WGRAD_STREAM
SPK_EVAL_STREAMS=1 bash tools/pmc_sq.sh ${TAG}_infer --mode infer > gpurun_out/${TAG}_sq.log 2>&1
python3 tools/pmc_sq_table.py ${TAG}_infer > gpurun_out/${TAG}_sq_counters_infer_calibrated.txt 2>> gpurun_out/${TAG}_sq.log || echo "sq table failed"
python3 tools/layer_roofs.py gpurun_out/${TAG}_infer_calibrated_layers.json > gpurun_out/${TAG}_infer_layer_roofs.txt
