mkdir -p gpurun_out/r4
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r4/test_all.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4/test_all.txt | tail -25
