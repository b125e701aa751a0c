#!/bin/bash
mkdir -p gpurun_out
export SPK_EVAL_STREAMS=1
bash tools/pmc_sq.sh r04_infer --mode infer > gpurun_out/r04_sq.log 2>&1
python3 tools/pmc_sq_table.py r04_infer gpurun_out/r04_sq_counters_infer_calibrated.txt | head -30
