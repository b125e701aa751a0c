#!/bin/bash
mkdir -p gpurun_out/r4final
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/r4final/test.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4final/test.txt | tail -8
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -3
