mkdir -p gpurun_out/r4
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_trained.py -q -s > gpurun_out/r4/test_k.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4/test_k.txt | grep -E "trained resnet18, batch|passed|failed|^E  " | tail -8
python3 bench.py --mode train --no-cpu-baseline --steps 3 --warmup 2 > /dev/null 2>&1
rm -rf gpurun_out/prof_k
SPK_WGRAD_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_k -- python3 bench.py --mode train --no-cpu-baseline --steps 6 --warmup 3 > /dev/null 2> gpurun_out/r4/prof_k.err
t=$(find gpurun_out/prof_k -name "*kernel_trace.csv" | head -1)
python3 tools/step_sequence.py "$t" 0 > gpurun_out/r4/train_sequence.txt
rm -rf gpurun_out/prof_k
wc -l gpurun_out/r4/train_sequence.txt
