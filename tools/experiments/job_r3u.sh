cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r3u
run() { timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); print('$1', 'march %.2f ms' % j['roofline']['avg_launch_ms'], 'step %.2f' % j['ms_per_step'], 'parity', j.get('parity_linf'), j.get('parity_linf_all'))
"; }
(
NB_MS6_PAIR=0 run unpaired
NB_MS6_PAIR=1 run pair_lag7
NB_MS6_PAIR=0 run unpaired
NB_MS6_PAIR=1 run pair_lag7
tools/experiments/abl_ms6.sh run
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "f16f6" 2>&1 | tail -3
) > gpurun_out/r3u/log.txt 2>&1
cat gpurun_out/r3u/log.txt
