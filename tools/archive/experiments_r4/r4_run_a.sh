mkdir -p gpurun_out/r4
timeout -k 10 900 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_train.py tests/test_gpu_calibrated.py -q > gpurun_out/r4/test_train.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4/test_train.txt | tail -12
for f in 1 0; do
  SPK_BNB_FUSE=$f timeout -k 10 200 python bench.py --mode train --no-cpu-baseline > gpurun_out/r4/bench_train_bnb$f.json 2>gpurun_out/r4/bench_train_bnb$f.err
  python -c "
import json; d=json.load(open('gpurun_out/r4/bench_train_bnb$f.json')); print('BNB_FUSE=$f', d['value'], d['ms_per_step'], d['roofline']['phases_ms'])"
done
for b in 512 768; do
  SPK_WGRAD_BLOCKS=$b timeout -k 10 200 python bench.py --mode train --no-cpu-baseline > gpurun_out/r4/bench_train_wb$b.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r4/bench_train_wb$b.json')); print('WGRAD_BLOCKS=$b', d['value'], d['ms_per_step'])"
done
