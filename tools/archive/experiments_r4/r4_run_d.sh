mkdir -p gpurun_out/r4
run() { echo "=== $*"; timeout -k 10 400 "$@" 2>&1 | grep -v amdgpu.ids | grep -E "trained|passed|failed|accuracy" | tail -6; }
run python -m pytest tests/test_gpu_trained.py -q -s -k resnet18
run env SPK_BNB_FUSE=0 python -m pytest tests/test_gpu_trained.py -q -s -k resnet18
run env SPK_WGRAD_BLOCKS=1024 python -m pytest tests/test_gpu_trained.py -q -s -k resnet18
run env SPK_TUNE_CACHE=off python -m pytest tests/test_gpu_trained.py -q -s -k resnet18
run python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_train.py tests/test_gpu_trained.py -q -s -k "resnet18 or dgrad or unfreeze"
