# round 5, job e: march A/B (base | M1 + M2: colour head prefetches the view phase's pieces, fold chunk MFMAs ahead of the refill), phase stamps
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5e; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_fold.py -x -q > $O/pytest_fold.txt 2>&1; tail -3 $O/pytest_fold.txt
for v in _base _m12 _base _m12; do echo "== variant '$v'" >> $O/time.log; NB_LIB_PATH=neuralbody_amd/lib/libnb_hip$v.so timeout 300 python tools/experiments/fold_check.py time >> $O/time.log 2>&1; done
grep -E "variant|march" $O/time.log
NB_LIB_PATH=neuralbody_amd/lib/libnb_hip_m12.so timeout 300 python tools/experiments/fold_check.py full small > $O/check_m12.log 2>&1; tail -4 $O/check_m12.log
NB_LIB_PATH=neuralbody_amd/lib/libnb_hip_m12t.so timeout 300 python tools/experiments/fold_phase_times.py > $O/phases_m12.md 2>&1; cat $O/phases_m12.md
