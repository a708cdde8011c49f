cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r3ab
run() { timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); print('$1', 'march %.2f ms' % j['roofline']['avg_launch_ms'], 'step %.2f' % j['ms_per_step'], 'rest %.3f' % (j['ms_per_step'] - j['roofline']['avg_launch_ms']), 'parity', j.get('parity_linf'))
"; }
(
for v in 0 1 0 1; do NB_CONV_LDS2=$v run lds2_$v; done
timeout 900 python -m pytest tests -q -x -m gpu -k "enc or golden or small or parity or novel" 2>&1 | tail -3
rocprofv3 --kernel-trace -d gpurun_out/r3ab/prof -o x -- python bench.py --steps 4 --warmup 2 --no-extras --no-cpu-baseline > /dev/null 2>&1
DB=$(find gpurun_out/r3ab/prof -name "*.db" | head -1)
python tools/rocpd_timeline.py $DB nb_march > gpurun_out/r3ab/step_timeline.md
grep -E "conv|from the end|latent" gpurun_out/r3ab/step_timeline.md | cut -c1-150
find gpurun_out -name "*.db" -delete
) > gpurun_out/r3ab/log.txt 2>&1
cat gpurun_out/r3ab/log.txt
