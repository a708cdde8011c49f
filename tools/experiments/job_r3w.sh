cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r3w
watch_run() {  # $1 tag, rest: env + command
  tag=$1; shift
  echo "=== $tag"
  ( "$@" > gpurun_out/r3w/bench_$tag.log 2>&1 ) &
  pid=$!
  sleep 14
  for i in 1 2 3 4 5 6; do
    timeout 20 rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power|mclk|fclk|socclk" | sed 's/ \+/ /g' | tr '\n' ';'; echo
    sleep 0.7
  done
  wait $pid
  grep '^{' gpurun_out/r3w/bench_$tag.log | python -c "
import sys, json
for line in sys.stdin:
    j = json.loads(line); print('$tag', 'march %.2f ms' % j['roofline']['avg_launch_ms'], 'step %.2f' % j['ms_per_step'])
"
}
(
echo "--- idle"; timeout 20 rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power" | sed 's/ \+/ /g' | tr '\n' ';'; echo
watch_run ms6_unpaired env NB_MS6_PAIR=0 python bench.py --no-cpu-baseline --no-extras --steps 700 --warmup 3 --precision f16f6
watch_run ms6_paired env NB_MS6_PAIR=1 python bench.py --no-cpu-baseline --no-extras --steps 700 --warmup 3 --precision f16f6
watch_run ring env python bench.py --no-cpu-baseline --no-extras --steps 700 --warmup 3 --precision f16f6r
watch_run f32 env python bench.py --no-cpu-baseline --no-extras --steps 150 --warmup 3 --precision f32
timeout 30 amd-smi metric -g 0 --power --clock 2>&1 | head -40
) > gpurun_out/r3w/log.txt 2>&1
cat gpurun_out/r3w/log.txt
