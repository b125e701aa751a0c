#!/bin/bash
# Per-kernel SQ counters (wave-cycle split, instruction mix, LDS bank conflicts) of a bench.py run.
# Usage (GPU box, repo root): bash tools/pmc_sq.sh <tag> <bench.py arguments...>; then python3 tools/pmc_sq_table.py <tag>
set -e
TAG=$1; shift
export TMPDIR=/tmp
export SPK_TUNE_CACHE=$PWD/gpurun_out/tune_sq_${TAG}.txt
python3 bench.py "$@" --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2>&1
i=0
for SET in "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rm -rf gpurun_out/pmc_sq_${TAG}_$i
  rocprofv3 --pmc $SET --output-format csv -d gpurun_out/pmc_sq_${TAG}_$i -- python3 bench.py "$@" --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> gpurun_out/pmc_sq_${TAG}_$i.err || echo "set $i failed"
  f=$(find gpurun_out/pmc_sq_${TAG}_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/${TAG}_pmc_sq_set$i.csv
done
ls -la gpurun_out/${TAG}_pmc_sq_set*.csv
