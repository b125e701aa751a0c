mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_effnet.py tests/test_gpu_fp8.py -q -x > gpurun_out/r4/test_b.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4/test_b.txt | tail -12
for f in 7 1 2 4 3 5 6 0; do
  SPK_BNB_FUSE=$f timeout -k 10 200 python bench.py --mode train --no-cpu-baseline > gpurun_out/r4/bench_train_bnbm$f.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r4/bench_train_bnbm$f.json')); p=d['roofline']['phases_ms']; print('BNB_FUSE=$f', d['value'], d['ms_per_step'], 'bn_bwd', p['bn_bwd (reduce+finalize+apply)'], 'dgrad', p['conv_dgrad (conv_igemm_kernel)'])"
done
for b in 256 384 512 1024; do
  SPK_WGRAD_BLOCKS=$b timeout -k 10 200 python bench.py --mode train --no-cpu-baseline > gpurun_out/r4/bench_train_wb$b.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r4/bench_train_wb$b.json')); print('WGRAD_BLOCKS=$b', d['value'], d['ms_per_step'])"
done
for net in efficientnet_b4; do for pr in mixed fp8; do for st in 2 1; do
  SPK_EVAL_STREAMS=$st timeout -k 10 300 python bench.py --network $net --batch 128 --precision $pr --mode infer --no-cpu-baseline > gpurun_out/r4/bench_${net}_${pr}_s$st.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r4/bench_${net}_${pr}_s$st.json')); print('$net $pr streams=$st', d['value'], d['ms_per_step'])"
done; done; done
