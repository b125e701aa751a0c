#!/bin/bash
# Round-3 evidence on one GPU box, in one call (run LAST, after the final source edit: the PMC traffic file is stamped
# with the conv-kernel source digest bench.py checks).  Usage (repo root): bash tools/evidence_r3.sh
#   1. warm the tuner cache; 2. bench lines (default, per-layer table); 3. rocprofv3 --kernel-trace --stats of the same
#   commands (warm cache); 4. PMC passes: FETCH_SIZE / WRITE_SIZE (infer and train), SQ MFMA-busy split.
set -e
TAG=r03
export TMPDIR=/tmp
export SPK_TUNE_CACHE=$PWD/gpurun_out/tune_${TAG}.txt
rm -f $SPK_TUNE_CACHE
python3 bench.py --mode both --no-cpu-baseline --steps 3 --warmup 2 > gpurun_out/${TAG}_warm.json 2> gpurun_out/${TAG}_warm.err
cp $SPK_TUNE_CACHE gpurun_out/${TAG}_tune_cache.txt
python3 bench.py --layers-out gpurun_out/${TAG}_infer_mixed_layers.json > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench_line.err
echo "bench line done"
# rocprofv3 / PMC passes run the eval forward on ONE stream (SPK_EVAL_STREAMS=1): per-kernel durations and counters are
# only a kernel's own when the two half-batch chains do not overlap; the bench line above is the two-stream default
export SPK_EVAL_STREAMS=1
for MODE in infer train; do
  rm -rf gpurun_out/prof_${TAG}_${MODE}
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_${MODE} -- python3 bench.py --mode $MODE --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/${TAG}_${MODE}_under_rocprof.json 2> gpurun_out/${TAG}_${MODE}_rocprof.err
  f=$(find gpurun_out/prof_${TAG}_${MODE} -name "*kernel_stats.csv" | head -1)
  cp "$f" gpurun_out/${TAG}_${MODE}_kernel_stats.csv
  rm -rf gpurun_out/prof_${TAG}_${MODE}
  echo "kernel trace $MODE done"
done
for MODE in infer train; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_${TAG}_${MODE}_$C
    rocprofv3 --pmc $C --output-format csv -d gpurun_out/pmc_${TAG}_${MODE}_$C -- python3 bench.py --mode $MODE --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> gpurun_out/pmc_${TAG}_${MODE}_$C.err
    echo "pmc $MODE $C done"
  done
  F=$(find gpurun_out/pmc_${TAG}_${MODE}_FETCH_SIZE -name "*counter_collection.csv" | head -1)
  W=$(find gpurun_out/pmc_${TAG}_${MODE}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
  NAME=$([ $MODE = infer ] && echo infer_mixed || echo train_bf16)
  python3 tools/pmc_traffic.py "$F" "$W" gpurun_out/${TAG}_pmc_traffic_${NAME}.json $MODE > gpurun_out/${TAG}_pmc_traffic_${MODE}.log
  rm -rf gpurun_out/pmc_${TAG}_${MODE}_FETCH_SIZE gpurun_out/pmc_${TAG}_${MODE}_WRITE_SIZE
done
rm -rf gpurun_out/pmc_${TAG}_sq
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc_${TAG}_sq -- python3 bench.py --mode infer --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> gpurun_out/pmc_${TAG}_sq.err
S=$(find gpurun_out/pmc_${TAG}_sq -name "*counter_collection.csv" | head -1)
python3 tools/pmc_mfma.py "$S" gpurun_out/${TAG}_pmc_mfma_util_infer_mixed.json > gpurun_out/${TAG}_pmc_mfma.log
rm -rf gpurun_out/pmc_${TAG}_sq
tail -n 4 gpurun_out/${TAG}_pmc_traffic_infer.log; tail -n 4 gpurun_out/${TAG}_pmc_mfma.log
