# round 5, job f: serial step timeline (encoder launches one by one), training step kernel stats
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5f; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace -d $O/tl -o t -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-overlap > $O/tl.log 2>&1
python tools/rocpd_timeline.py $(find $O/tl -name "*.db" | head -1) > $O/step_timeline.md 2>&1; tail -3 $O/step_timeline.md
timeout 600 rocprofv3 --kernel-trace -d $O/tr -o t -- python bench.py --mode train --steps 10 --warmup 3 > $O/tr.log 2>&1
python tools/rocpd_summary.py $(find $O/tr -name "*.db" | head -1) > $O/train_kernel_stats.md 2>&1; tail -2 $O/train_kernel_stats.md
find $O -name "*.db" -delete
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
