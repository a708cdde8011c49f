cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r3o
run() { # tag, lib, precision, size, samples
  NB_LIB_PATH=$2 timeout 300 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline --precision $3 --size $4 --samples $5 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); print('$1', 'march %.2f ms' % j['roofline']['avg_launch_ms'], 'step %.2f' % j['ms_per_step'], 'parity', j.get('parity_linf'))
"
}
L=$PWD/neuralbody_amd/lib/libnb_hip.so; N=$PWD/neuralbody_amd/lib/libnb_hip_noslp.so
run ms6_512 $L f16f6 512 64
run ring_512 $L f16f6r 512 64
run ring_noslp_512 $N f16f6r 512 64
run f8_512 $L f16f8 512 64
run f8_noslp_512 $N f16f8 512 64
run ms6_512b $L f16f6 512 64
run ring_512b $L f16f6r 512 64
run ms6_1024 $L f16f6 1024 128
run ring_1024 $L f16f6r 1024 128
