cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "beyond_fp16 or small_eval or trained" -s 2>&1 | grep -v "^$" | tail -15
for p in f16f6 f16f6r; do timeout 200 python bench.py --no-cpu-baseline --no-extras --precision $p 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); print('$p', j['roofline']['kernel'], 'march %.2f ms' % j['roofline']['avg_launch_ms'], 'step %.2f' % j['ms_per_step'], 'parity', j.get('parity_linf'), j.get('parity_linf_all'))
"; done
