# round 6, job n: the serial step timeline (every launch between two marches) on the final tree
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6n; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace -d $O/tl -o t -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-overlap > $O/tl.log 2>&1
python tools/rocpd_timeline.py $(find $O/tl -name "*.db" | head -1) > $O/step_timeline.md 2>&1; tail -2 $O/step_timeline.md
find $O -name "*.db" -delete
