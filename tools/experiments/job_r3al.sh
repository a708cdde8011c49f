cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3al; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; grep -E "passed|failed" $O/gpu_tests.log | tail -1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json
rocprofv3 --kernel-trace -d $O/prof -o x -- python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_prof.log 2>&1
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_summary.py $DB > $O/kernel_stats.md 2>/dev/null
python tools/rocpd_timeline.py $DB nb_march > $O/step_timeline.md 2>/dev/null
sed -n 5,9p $O/kernel_stats.md | cut -c1-160; tail -1 $O/step_timeline.md
find gpurun_out -name "*.db" -delete
