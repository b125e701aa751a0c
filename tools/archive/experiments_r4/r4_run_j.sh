mkdir -p gpurun_out/r4
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r4/test_all4.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4/test_all4.txt | tail -15
for i in 1 2; do timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r4/bench_both_$i.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r4/bench_both_$i.json')); print('infer', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_frac'], 'train', d['train']['value'], d['train']['ms_per_step'])"; done
