mkdir -p gpurun_out/r4
timeout -k 10 1100 python -m pytest tests/test_gpu_train.py tests/test_gpu_effnet_train.py tests/test_gpu_fp8.py tests/test_gpu_trained.py tests/test_gpu_calibrated.py tests/test_gpu_infer.py tests/test_gpu_effnet.py -q -s > gpurun_out/r4/test_i.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4/test_i.txt | grep -E "max \|dp\||passed|failed|FAILED|^E  |trained|worst \|GPU|epoch 3 base.0|base 1.3" | tail -70
