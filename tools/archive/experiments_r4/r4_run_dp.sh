#!/bin/bash
mkdir -p gpurun_out/r4dp2
timeout -k 10 900 python -m pytest tests/test_gpu_dp.py -m gpu -q -x > gpurun_out/r4dp2/test.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4dp2/test.txt | tail -30
