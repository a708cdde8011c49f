cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4y; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -o s -- python bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-extras > $O/stats.log 2>&1
python tools/rocpd_summary.py $(find $O/stats -name "*.db") > $O/kernel_stats.md 2>&1
find $O -name "*.db" -delete
bash tools/pmc_march.sh f16f6 r4y/pmc > $O/pmc_out.txt 2>&1
bash tools/pmc_traffic.sh f16f6 r4y/traffic > $O/traffic_out.txt 2>&1
python bench.py --steps 1500 --warmup 3 --no-cpu-baseline --no-extras > $O/loop.log 2>&1 &
sleep 14; for i in 1 2 3; do rocm-smi --showpower --showclocks >> $O/smi_load.txt 2>&1; sleep 2; done; wait
grep -i "sclk\|power (W)" $O/smi_load.txt | head -8; tail -1 $O/loop.log | cut -c1-300
