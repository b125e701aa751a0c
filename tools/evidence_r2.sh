#!/bin/bash
# Round-2 evidence, one GPU box, repo root: warm-cache rocprofv3 kernel stats (infer / train), PMC traffic + MFMA
# utilisation, the bench lines of every BASELINE config that fits one GPU, per-kernel SQ counter tables, end-to-end
# rates.  Everything lands in gpurun_out/r02_*; copy what is to be judged into profiles/.
export TMPDIR=/tmp
T=r02
bash tools/prof_r2.sh $T > gpurun_out/${T}_prof.log 2>&1; echo "prof rc=$?"
bash tools/pmc_r2.sh $T > gpurun_out/${T}_pmc.log 2>&1; echo "pmc rc=$?"
# the default bench line quotes roofline.traffic from profiles/ (stamped with the kernel-source digest): put the fresh one there
cp gpurun_out/${T}_pmc_traffic_infer_mixed.json profiles/ 2>/dev/null
python3 bench.py > gpurun_out/${T}_bench_line.json 2> gpurun_out/${T}_bench_line.err; echo "bench default rc=$?"
export SPK_TUNE_CACHE=$PWD/gpurun_out/tune_${T}.txt
for MODE in mixed precise fast; do
  python3 bench.py --mode infer --precision $MODE --no-cpu-baseline --layers-out gpurun_out/${T}_infer_${MODE}_layers.json > gpurun_out/${T}_bench_infer_${MODE}.json 2>/dev/null; echo "infer $MODE rc=$?"
done
python3 bench.py --mode train --no-cpu-baseline --layers-out gpurun_out/${T}_train_phases.json > gpurun_out/${T}_bench_train.json 2>/dev/null; echo "train rc=$?"
python3 bench.py --network resnet18 --batch 512 > gpurun_out/${T}_bench_line_resnet18_b512.json 2>/dev/null; echo "r18 rc=$?"
unset SPK_TUNE_CACHE
for P in fp8 mixed; do for B in 128 256; do
  python3 bench.py --network efficientnet_b4 --batch $B --precision $P --mode infer --cpu-seconds 10 --layers-out gpurun_out/${T}_b4_${P}_b${B}_layers.json > gpurun_out/${T}_bench_line_efficientnet_b4_${P}_b${B}.json 2>/dev/null; echo "b4 $P $B rc=$?"
done; done
bash tools/pmc_sq.sh ${T}_r50 --mode both > gpurun_out/${T}_pmc_sq_r50.log 2>&1 && python3 tools/pmc_sq_table.py ${T}_r50 gpurun_out/${T}_sq_counters_resnet50.txt > /dev/null; echo "sq r50 rc=$?"
bash tools/pmc_sq.sh ${T}_b4 --network efficientnet_b4 --batch 256 --precision fp8 --mode infer > gpurun_out/${T}_pmc_sq_b4.log 2>&1 && python3 tools/pmc_sq_table.py ${T}_b4 gpurun_out/${T}_sq_counters_efficientnet_b4_fp8.txt > /dev/null; echo "sq b4 rc=$?"
timeout -k 10 300 python3 tools/e2e_prob.py 20000 resnet18 > gpurun_out/${T}_e2e_prob.log 2>&1; echo "e2e prob rc=$?"
timeout -k 10 300 python3 tools/e2e_train.py > gpurun_out/${T}_e2e_train.log 2>&1; echo "e2e train rc=$?"
tail -n 3 gpurun_out/${T}_e2e_prob.log; tail -n 3 gpurun_out/${T}_e2e_train.log
# gpurun merges at most 64 MiB back: drop the raw traces, keep the summaries
rm -rf gpurun_out/pmc_* gpurun_out/prof_* gpurun_out/*_pmc_sq_set*.csv
du -sh gpurun_out
