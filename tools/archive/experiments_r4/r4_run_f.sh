mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_train.py tests/test_gpu_trained.py tests/test_gpu_dp.py -q > gpurun_out/r4/test_f.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4/test_f.txt | tail -6
for d in 1 0 1 0; do
  SPK_BNB_DEFER=$d timeout -k 10 200 python bench.py --mode train --no-cpu-baseline > gpurun_out/r4/bench_train_defer$d.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r4/bench_train_defer$d.json')); p=d['roofline']['phases_ms']; print('DEFER=$d', d['value'], d['ms_per_step'], 'bn_bwd', p['bn_bwd (reduce+finalize+apply)'], 'dgrad', p['conv_dgrad (conv_igemm_kernel)'])"
done
