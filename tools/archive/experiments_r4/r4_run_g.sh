mkdir -p gpurun_out/r4
timeout -k 10 600 python -m pytest tests/test_gpu_trained.py -q -s > gpurun_out/r4/test_g.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4/test_g.txt | grep -E "trained|max \|dp\||passed|failed" | tail -20
timeout -k 10 200 python bench.py --network resnet18 --batch 512 --no-cpu-baseline > gpurun_out/r4/bench_r18_b512.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r4/bench_r18_b512.json')); print('resnet18 b512', d['config']['precision'], d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_frac'], 'train', d['train']['value'], d['train']['ms_per_step'])"
timeout -k 10 200 python bench.py --network resnet18 --batch 512 --mode infer --precision mixed --no-cpu-baseline > gpurun_out/r4/bench_r18_b512_mixed.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r4/bench_r18_b512_mixed.json')); print('resnet18 b512 mixed', d['value'], d['ms_per_step'])"
