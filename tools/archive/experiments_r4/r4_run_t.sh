#!/bin/bash
mkdir -p gpurun_out/r4t
timeout -k 10 900 python -m pytest tests/test_gpu_effnet.py -m gpu -q -k "squeeze" > gpurun_out/r4t/test.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4t/test.txt | tail -12
