cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r3x
export NB_MS6_PAIR=0
watch_run() {
  tag=$1; shift
  ( "$@" > gpurun_out/r3x/bench_$tag.log 2>&1 ) &
  pid=$!
  sleep 9
  for i in 1 2 3; do
    timeout 20 rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power \(W\)" | sed 's/.*(\([0-9]*\)Mhz).*/sclk \1/; s/.*Power (W): /W /' | tr '\n' ' '; echo
    sleep 0.5
  done
  wait $pid
  grep '^{' gpurun_out/r3x/bench_$tag.log | python -c "
import sys, json
for line in sys.stdin:
    j = json.loads(line); print('$tag', 'march %.2f ms' % j['roofline']['avg_launch_ms'], 'step %.2f' % j['ms_per_step'])
"
}
(
for so in neuralbody_amd/lib/libnb_hip_ms6*.so; do
  tag=$(basename $so .so | sed 's/libnb_hip_ms6//')
  NB_LIB_PATH=$PWD/$so watch_run $tag python bench.py --no-cpu-baseline --no-extras --steps 600 --warmup 3 --precision f16f6
done
) > gpurun_out/r3x/log.txt 2>&1
cat gpurun_out/r3x/log.txt
