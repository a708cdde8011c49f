cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r3n
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "f16f6" > gpurun_out/r3n/parity.log 2>&1
tail -4 gpurun_out/r3n/parity.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r3n/bench_ms6.json 2> gpurun_out/r3n/bench_ms6.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r3n/bench_ms6.json').read().strip().splitlines()[-1])
print('ms6 march %.2f ms step %.2f parity %s' % (j['roofline']['avg_launch_ms'], j['ms_per_step'], j.get('parity_linf')))
PY
tools/experiments/abl_ms6.sh run
timeout 300 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --precision f16f6r 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); print('ring', 'march %.2f ms' % j['roofline']['avg_launch_ms'], 'step %.2f' % j['ms_per_step'])
"
NB_LIB_PATH=$PWD/neuralbody_amd/lib/libnb_hip_ms6TIMING.so python tools/experiments/ms6_phase_times.py > gpurun_out/r3n/phases_2wg.log 2>&1
cat gpurun_out/r3n/phases_2wg.log
