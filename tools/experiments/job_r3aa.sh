cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r3aa
run() { timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); print('$1', 'march %.2f ms' % j['roofline']['avg_launch_ms'], 'step %.2f' % j['ms_per_step'], 'rest %.3f' % (j['ms_per_step'] - j['roofline']['avg_launch_ms']), 'parity', j.get('parity_linf'))
"; }
(
for ks in 0 8 4 0 8 4; do NB_CONV_KS=$ks run ks$ks; done
for ks in 8 4; do echo "tests ks=$ks"; NB_CONV_KS=$ks timeout 900 python -m pytest tests -q -x -m gpu -k "enc or golden or small or parity" 2>&1 | tail -3; done
NB_CONV_KS=8 rocprofv3 --kernel-trace -d gpurun_out/r3aa/prof -o x -- python bench.py --steps 4 --warmup 2 --no-extras --no-cpu-baseline > /dev/null 2>&1
DB=$(find gpurun_out/r3aa/prof -name "*.db" | head -1)
python tools/rocpd_timeline.py $DB nb_march > gpurun_out/r3aa/step_timeline_ks8.md
grep -E "conv|from the end" gpurun_out/r3aa/step_timeline_ks8.md | cut -c1-150
find gpurun_out -name "*.db" -delete
) > gpurun_out/r3aa/log.txt 2>&1
cat gpurun_out/r3aa/log.txt
