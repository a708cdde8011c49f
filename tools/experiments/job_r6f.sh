# round 6, job f: fp4 cross-term weights (NB_FP4_TERMS=3) first light: parity tests + march time
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6f; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fold.py tests/test_gpu_fullsize.py tests/test_gpu_fixup.py -q -m gpu -s -x > $O/tests.log 2>&1
echo "rc=$?" >> $O/tests.log
NB_LAST_SAMPLE_FIXUP=0 python tools/experiments/fold_check.py time > $O/time.log 2>&1
python bench.py > $O/bench.json 2> $O/bench.err
grep -E "L-inf|passed|failed|rc=" $O/tests.log | tail -40; cat $O/time.log | grep march
