cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -q -x -m gpu -k "auto" -s 2>&1 | grep -v "^$" | tail -8
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); print(j['roofline']['kernel'], 'march %.2f ms' % j['roofline']['avg_launch_ms'], 'step %.2f' % j['ms_per_step'], 'parity', j.get('parity_linf'), j.get('parity_linf_all'), j['dtype'])
"; done
NB_AUTO_TUNE=0 timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); print('no tune:', j['roofline']['kernel'], 'march %.2f ms' % j['roofline']['avg_launch_ms'], 'step %.2f' % j['ms_per_step'])
"
