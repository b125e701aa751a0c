mkdir -p gpurun_out/r4
rm -f ~/.cache/sykepic_hip/tune-*.txt
SPK_TUNE_LOG=1 timeout -k 10 1100 python -m pytest tests -m gpu -q -s -x --deselect tests/test_gpu_trained.py::test_trained_efficientnet_b0_fp16_calibrated_and_fp8 > gpurun_out/r4/test_all2.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4/test_all2.txt | grep -E "trained 300|passed|failed|FAILED" | tail -8
cp ~/.cache/sykepic_hip/tune-*.txt gpurun_out/r4/tune_after_suite.txt 2>/dev/null
echo "--- again, same cache, trained test alone"
timeout -k 10 300 python -m pytest tests/test_gpu_trained.py -q -s -k resnet18 2>&1 | grep -E "trained 300|passed|failed" | tail -3
