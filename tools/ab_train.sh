# scratch: training A/B
set -e
export TMPDIR=/tmp
export SPK_TUNE_CACHE=$PWD/gpurun_out/tune_t1.txt
python3 -m pytest tests/test_gpu_train.py -x -q -m gpu -k "launch_forms or reproducible or benched_size" > gpurun_out/t4_tests.txt 2>&1 || { tail -30 gpurun_out/t4_tests.txt; exit 1; }
tail -3 gpurun_out/t4_tests.txt
python3 bench.py --mode train --no-cpu-baseline --steps 3 --warmup 2 > /dev/null 2>&1
B="python3 bench.py --mode train --no-cpu-baseline --steps 40 --warmup 10"
for rep in 1 2 3; do
for v in "X=1" "SPK_EVENT_ON_KERNEL=0" "SPK_TAIL_DEFER=0"; do
  ( export $v; $B 2>> gpurun_out/t4.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['step_ms'])" )
done; done
trace() {
  local NAME=$1; shift
  rm -rf gpurun_out/prof_$NAME
  ( for kv in "$@"; do export "$kv"; done
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$NAME -- python3 bench.py --mode train --no-cpu-baseline --no-kernel-profile --steps 20 --warmup 5 > gpurun_out/t4_$NAME.json 2> gpurun_out/t4_$NAME.err )
  local t=$(find gpurun_out/prof_$NAME -name "*kernel_trace.csv" | head -1)
  python3 tools/step_timeline.py "$t" 60 > gpurun_out/t4_timeline_$NAME.txt
  rm -rf gpurun_out/prof_$NAME
}
trace new X=1
trace evmarker SPK_EVENT_ON_KERNEL=0
