#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and MFMA utilisation of the conv kernels of one steady-state
# ResNet-50 forward, with a warm tuner cache.  Usage (GPU box, repo root): bash tools/pmc_r2.sh <tag>
set -e
TAG=${1:-r02}
export TMPDIR=/tmp
export SPK_TUNE_CACHE=$PWD/gpurun_out/tune_${TAG}.txt
[ -f "$SPK_TUNE_CACHE" ] || python3 bench.py --mode infer --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_${TAG}_$C
  rocprofv3 --pmc $C --output-format csv -d gpurun_out/pmc_${TAG}_$C -- python3 bench.py --mode infer --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> gpurun_out/pmc_${TAG}_$C.err
done
F=$(find gpurun_out/pmc_${TAG}_FETCH_SIZE -name "*counter_collection.csv" | head -1)
W=$(find gpurun_out/pmc_${TAG}_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python3 tools/pmc_traffic.py "$F" "$W" gpurun_out/${TAG}_pmc_traffic_infer_mixed.json > gpurun_out/${TAG}_pmc_traffic.log
rm -rf gpurun_out/pmc_${TAG}_sq
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc_${TAG}_sq -- python3 bench.py --mode infer --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> gpurun_out/pmc_${TAG}_sq.err
S=$(find gpurun_out/pmc_${TAG}_sq -name "*counter_collection.csv" | head -1)
python3 tools/pmc_mfma.py "$S" gpurun_out/${TAG}_pmc_mfma_util_infer_mixed.json > gpurun_out/${TAG}_pmc_mfma.log
tail -n 5 gpurun_out/${TAG}_pmc_traffic.log; tail -n 5 gpurun_out/${TAG}_pmc_mfma.log
