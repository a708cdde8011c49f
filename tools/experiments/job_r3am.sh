cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_dist.py -q -x -m gpu -k "auto or dist or bench" -s 2>&1 | grep -E "measured|passed|failed" | tail -5
timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); print(j['roofline']['kernel'], 'march %.2f ms' % j['roofline']['avg_launch_ms'], 'step %.2f' % j['ms_per_step'], j['config'].get('auto_organisation'))
"
