# round 5, job l: what the convolutions' fp64 statistics atomics cost (a build without them: wrong numbers, timing only)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5l; mkdir -p $O
for v in "" _nostats "" _nostats; do echo "== variant '$v'" >> $O/time.log; NB_BENCH_PRECISION=f16f6 NB_LIB_PATH=neuralbody_amd/lib/libnb_hip$v.so timeout 300 python tools/experiments/encoder_time.py >> $O/time.log 2>&1; done
grep -E "variant|encoder" $O/time.log
