cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r3z
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r3z/gpu_tests.log 2>&1
tail -4 gpurun_out/r3z/gpu_tests.log; grep -n "FAILED\|Error" gpurun_out/r3z/gpu_tests.log | head
timeout 600 python bench.py > gpurun_out/r3z/bench.json 2> gpurun_out/r3z/bench.err
python - <<'PY'
import json
for line in open('gpurun_out/r3z/bench.json'):
    if line.startswith('{'):
        j = json.loads(line)
        print({k: j[k] for k in ('value', 'ms_per_step', 'median_ms_per_step', 'parity_linf', 'parity_linf_all', 'dtype')})
        print(j['roofline']); print(j.get('extras')); print(j['cpu_baseline'])
PY
tail -3 gpurun_out/r3z/bench.err
