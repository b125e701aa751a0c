mkdir -p gpurun_out/r4
python tests/archive/diagnostics/input_rounding.py 2>&1 | grep -v amdgpu.ids | tail -4
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r4/test_all3.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4/test_all3.txt | tail -30
