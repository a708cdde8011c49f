# round 5, job g: conv16_lds prefetch depth 2 / 3 / 4 (encoder alone, training step), encoder parity with the deepest
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5g; mkdir -p $O
for v in "" _cd3 _cd4 "" _cd3 _cd4; do echo "== variant '$v'" >> $O/time.log; NB_LIB_PATH=neuralbody_amd/lib/libnb_hip$v.so timeout 300 python tools/experiments/encoder_time.py train >> $O/time.log 2>&1; done
grep -E "variant|encoder|train" $O/time.log
NB_LIB_PATH=neuralbody_amd/lib/libnb_hip_cd4.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_backward.py -x -q -k "encoder or train or backward or gradient" > $O/pytest_cd4.txt 2>&1; tail -3 $O/pytest_cd4.txt
timeout 600 python -m pytest tests/test_gpu_frames.py -x -q > $O/pytest_frames.txt 2>&1; tail -2 $O/pytest_frames.txt
