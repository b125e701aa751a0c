#!/bin/bash
# Regenerates syke-pic_amd/sykepic_hip/tune_seed_gfx950.txt on one MI355X: every tuner times its candidates from an
# empty cache for the shapes of the shipped benchmarks (ResNet-50 inference in the calibrated and the mixed mode + the
# training step at batch 256, ResNet-18 both, EfficientNet-B0 / B4 inference at batch 128), the winners are written sorted
# under the file's header.  Usage (repo root, on the GPU box): bash tools/make_tune_seed.sh ; the result lands in
# gpurun_out/tune_seed_gfx950.txt (copy it over the tracked file).
set -e
mkdir -p gpurun_out
export SPK_TUNE_SEED=0
export SPK_TUNE_CACHE=$PWD/gpurun_out/tune_seed_raw.txt
rm -f $SPK_TUNE_CACHE
B="python3 bench.py --no-cpu-baseline --steps 3 --warmup 2"
$B > gpurun_out/seed_r50.json 2> gpurun_out/seed.err
$B --network resnet18 > gpurun_out/seed_r18.json 2>> gpurun_out/seed.err
$B --mode infer --network efficientnet_b0 --batch 128 > gpurun_out/seed_b0.json 2>> gpurun_out/seed.err
$B --mode infer --network efficientnet_b4 --batch 128 > gpurun_out/seed_b4.json 2>> gpurun_out/seed.err
$B --mode infer --network efficientnet_b4 --batch 128 --precision fp8 > gpurun_out/seed_b4f8.json 2>> gpurun_out/seed.err || true
{
  grep "^#" syke-pic_amd/sykepic_hip/tune_seed_gfx950.txt
  grep -v "^#" $SPK_TUNE_CACHE | sort -u
} > gpurun_out/tune_seed_gfx950.txt
wc -l gpurun_out/tune_seed_gfx950.txt
