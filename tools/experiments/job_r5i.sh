# round 5, job i: the 128-sample organisation of the march (NB_MARCH_NU=2): parity, time against the shipped kernel
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5i; mkdir -p $O
NB_MARCH_NU=2 timeout 600 python tools/experiments/fold_check.py small full > $O/check_nu2.log 2>&1; tail -6 $O/check_nu2.log
for nu in 1 2 1 2; do echo "== NB_MARCH_NU=$nu" >> $O/time.log; NB_MARCH_NU=$nu timeout 300 python tools/experiments/fold_check.py time >> $O/time.log 2>&1; done
grep -E "NU=|march" $O/time.log
NB_MARCH_NU=2 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fold.py tests/test_gpu_fullsize.py tests/test_gpu_frames.py -x -q > $O/pytest_nu2.txt 2>&1; tail -4 $O/pytest_nu2.txt
