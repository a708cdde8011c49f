cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r3ad
run() { timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); print('$1', 'march %.2f ms' % j['roofline']['avg_launch_ms'], 'step %.2f' % j['ms_per_step'], 'rest %.3f' % (j['ms_per_step'] - j['roofline']['avg_launch_ms']), 'parity', j.get('parity_linf'))
"; }
(
for v in 512 100000 512 100000; do NB_CONV_LDS2_MAX=$v run lds2max_$v; done
NB_CONV_LDS2=0 run lds2_off
) > gpurun_out/r3ad/log.txt 2>&1
cat gpurun_out/r3ad/log.txt
