set -e
export TMPDIR=/tmp
mkdir -p gpurun_out
python3 bench.py --mode infer --network efficientnet_b4 --batch 128 --no-cpu-baseline --steps 3 --warmup 2 > /dev/null 2>&1
rm -rf gpurun_out/prof_b4
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b4 -- python3 bench.py --mode infer --network efficientnet_b4 --batch 128 --no-cpu-baseline --no-kernel-profile --steps 20 --warmup 5 > gpurun_out/b4_rocprof.json 2> gpurun_out/b4_rocprof.err
t=$(find gpurun_out/prof_b4 -name "*kernel_trace.csv" | head -1)
python3 tools/step_timeline.py "$t" 40 > gpurun_out/b4_timeline.txt
rm -rf gpurun_out/prof_b4
head -60 gpurun_out/b4_timeline.txt
