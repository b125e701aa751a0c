#!/bin/bash
# Round-3 evidence for the EfficientNet training step and the per-stream timelines, one GPU box, one call (repo root):
#   bash tools/evidence_r3_effnet.sh
# bench lines (B0 / B4 training), rocprofv3 kernel stats of the B0 run, per-stream timelines of one steady-state step
# (tools/step_timeline.py) for B0, B4 and ResNet-50 training, and the batch-scaling probe (tools/launch_bound_probe.py).
set -e
TAG=r03
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
for cfg in "0 128" "4 128" "4 64"; do
  set -- $cfg
  python3 bench.py --network efficientnet_b$1 --mode train --batch $2 --no-cpu-baseline 2> /dev/null | tail -1 > $OUT/${TAG}_bench_effnet_b$1_train_b$2.json
  echo "bench b$1 batch $2 done"
done
D=/tmp/kt_stats_$$
rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 bench.py --network efficientnet_b0 --mode train --batch 128 --no-cpu-baseline --steps 20 --warmup 5 > /dev/null 2>&1
cp "$(find $D -name '*kernel_stats.csv' | head -1)" $OUT/${TAG}_effnet_b0_train_kernel_stats.csv
for cfg in "efficientnet_b0 128 effnet_b0_train" "efficientnet_b4 128 effnet_b4_train" "resnet50 256 resnet50_train"; do
  set -- $cfg
  D=/tmp/kt_$3_$$
  rocprofv3 --kernel-trace --output-format csv -d $D -- python3 tools/launch_bound_probe.py $1 $2 > /dev/null 2>&1
  python3 tools/step_timeline.py "$(find $D -name '*kernel_trace.csv' | head -1)" 24 > $OUT/${TAG}_timeline_$3.txt
  echo "timeline $3 done"
done
for cfg in "efficientnet_b0 32" "efficientnet_b0 128" "efficientnet_b4 64" "efficientnet_b4 128" "resnet50 256"; do
  python3 tools/launch_bound_probe.py $cfg 2> /dev/null | tail -1
done > $OUT/${TAG}_launch_bound_probe.txt
cat $OUT/${TAG}_launch_bound_probe.txt
