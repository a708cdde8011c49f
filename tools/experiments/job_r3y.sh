cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r3y
( tools/experiments/abl_ms6.sh run; tools/experiments/abl_ms6.sh run
for so in neuralbody_amd/lib/libnb_hip_ms6*.so; do echo $so; NB_LIB_PATH=$PWD/$so timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "f16f6 and not f16f6r" 2>&1 | tail -2; done
NB_LIB_PATH=$PWD/neuralbody_amd/lib/libnb_hip_ms6base.so timeout 200 python bench.py --no-cpu-baseline --no-extras --precision f16f6r | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); print('ring', 'march %.2f ms' % j['roofline']['avg_launch_ms'], 'step %.2f' % j['ms_per_step'], 'parity', j.get('parity_linf'))
"
) > gpurun_out/r3y/log.txt 2>&1
cat gpurun_out/r3y/log.txt
