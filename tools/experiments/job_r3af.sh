cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3af; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; tail -2 $O/gpu_tests.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
# kernel stats + timeline of the default bench line without the extras legs
rocprofv3 --kernel-trace -d $O/prof -o x -- python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_prof.log 2>&1
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_summary.py $DB > $O/kernel_stats.md 2>/dev/null
python tools/rocpd_timeline.py $DB nb_march > $O/step_timeline.md 2>/dev/null
head -12 $O/kernel_stats.md | cut -c1-200
find gpurun_out -name "*.db" -delete
bash tools/pmc_march.sh f16f6r r3af_pmc > /dev/null 2>&1; cp gpurun_out/r3af_pmc_summary.txt $O/pmc_march_raw.txt
bash tools/pmc_traffic.sh f16f6r r3af_traffic > /dev/null 2>&1; cp gpurun_out/r3af_traffic_summary.txt $O/traffic_raw.txt
cat $O/pmc_march_raw.txt $O/traffic_raw.txt
rm -rf gpurun_out/r3af_pmc_* gpurun_out/r3af_traffic_*
