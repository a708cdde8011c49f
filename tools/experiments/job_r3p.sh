cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r3p
timeout 1200 python -m pytest tests -q -m gpu > gpurun_out/r3p/gpu_tests.log 2>&1
tail -6 gpurun_out/r3p/gpu_tests.log
grep -h "trained/\|ill-conditioned\|rgb L-inf of" gpurun_out/r3p/gpu_tests.log | head -20
timeout 600 python bench.py --steps 10 --warmup 3 --no-extras > gpurun_out/r3p/bench.json 2> gpurun_out/r3p/bench.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r3p/bench.json').read().strip().splitlines()[-1])
print('march %.2f ms step %.2f' % (j['roofline']['avg_launch_ms'], j['ms_per_step']))
print(json.dumps(j.get('parity'), indent=0)[:1200])
print(j.get('cpu_baseline'))
PY
