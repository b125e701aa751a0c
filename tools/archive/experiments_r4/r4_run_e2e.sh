#!/bin/bash
mkdir -p gpurun_out/r4e2e
timeout -k 10 500 python tools/e2e_prob.py 20000 resnet18 > gpurun_out/r4e2e/mixed.txt 2> gpurun_out/r4e2e/mixed.err
tail -2 gpurun_out/r4e2e/mixed.txt
E2E_CALIBRATE=1 timeout -k 10 500 python tools/e2e_prob.py 20000 resnet18 > gpurun_out/r4e2e/calibrated.txt 2> gpurun_out/r4e2e/calibrated.err
tail -3 gpurun_out/r4e2e/calibrated.txt
