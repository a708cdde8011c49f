# round 5, job p: fold planes with 64-row workgroups on the 128-channel levels: fold tests, timeline
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fold.py tests/test_gpu_parity.py -x -q > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -2
timeout 600 rocprofv3 --kernel-trace -d $O/tl -o t -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-overlap > $O/tl.log 2>&1
python tools/rocpd_timeline.py $(find $O/tl -name "*.db" | head -1) > $O/step_timeline.md 2>&1; tail -2 $O/step_timeline.md; grep fold_rows $O/step_timeline.md
find $O -name "*.db" -delete
