#!/bin/bash
# The counter half of tools/evidence_r5.sh alone (its first run of the final sources stopped in tools/pmc_traffic.py: the
# step marks were matched on a kernel name that the input conversion no longer has).  Same box = same call as nothing else.
set -e
TAG=r05
export TMPDIR=/tmp
mkdir -p gpurun_out
export SPK_TUNE_CACHE=$PWD/gpurun_out/tune_${TAG}.txt
rm -f $SPK_TUNE_CACHE
# start from the shipped tuner seed - what a fresh box (the driver's) starts from - and tune whatever it does not list
grep -v "^#" syke-pic_amd/sykepic_hip/tune_seed_gfx950.txt > $SPK_TUNE_CACHE
python3 bench.py --mode both --no-cpu-baseline --steps 3 --warmup 2 > gpurun_out/${TAG}_warm.json 2> gpurun_out/${TAG}_warm.err
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
# round 5 extras: the whole-bottleneck kernel alone (both block forms, phase stamps, memory ablations) and `sykepic prob` end to end
# on a model directory as the reference leaves it (no act_means.pth: calibrated on its first batch) / calibrated beforehand
unset SPK_EVAL_STREAMS SPK_WGRAD_STREAM
( for F in 0 4; do echo "== SPK_BNECK_FLAGS=$F (0: 14-row blocks of 8 waves, 4: 7-row blocks of 4 waves)"; SPK_BNECK_FLAGS=$F STAMPS=1 python3 tools/bneck_bench.py 256 128 64; done ) > gpurun_out/${TAG}_bneck_bench.txt 2>&1
( for F in 0 8 16 32 48; do echo "== SPK_BNECK_FLAGS=$F (8: phase 1 reads one image, 16: no shortcut loads, 32: no stores)"; SPK_BNECK_FLAGS=$F STAMPS=1 python3 tools/bneck_bench.py 128 2>&1 | grep -A2 "stage[23]"; done ) > gpurun_out/${TAG}_bneck_ablation.txt 2>&1
( echo "== model directory as the reference leaves it (auto-calibration on the first batch)"; python3 tools/e2e_prob.py 20000 resnet18; echo "== sykepic calibrate first"; E2E_CALIBRATE=1 python3 tools/e2e_prob.py 20000 resnet18 ) > gpurun_out/${TAG}_e2e_prob.txt 2>&1
tail -n 3 gpurun_out/${TAG}_e2e_prob.txt
