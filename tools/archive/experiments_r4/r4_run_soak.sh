#!/bin/bash
mkdir -p gpurun_out/r4soak
timeout -k 10 900 python tools/soak_streams.py 100 > gpurun_out/r4soak/soak.txt 2> gpurun_out/r4soak/soak.err
cat gpurun_out/r4soak/soak.txt; tail -3 gpurun_out/r4soak/soak.err
