# round 6, job m: the records of the round on the final tree — GPU tests + smoke, default bench, kernel stats, serial step timeline,
# training step, PMC passes, HBM traffic, power / clocks under load, full-view parity, configs 3 and 5
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6m; mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu -s > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -o s -- python bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-extras > $O/stats.log 2>&1
python tools/rocpd_summary.py $(find $O/stats -name "*.db" | head -1) > $O/kernel_stats.md 2>&1
timeout 600 rocprofv3 --kernel-trace -d $O/tl -o t -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-overlap > $O/tl.log 2>&1
python tools/rocpd_timeline.py $(find $O/tl -name "*.db" | head -1) > $O/step_timeline.md 2>&1; tail -2 $O/step_timeline.md
timeout 600 rocprofv3 --kernel-trace -d $O/tr -o t -- python bench.py --mode train --steps 10 --warmup 3 > $O/tr.log 2>&1
python tools/rocpd_summary.py $(find $O/tr -name "*.db" | head -1) > $O/train_kernel_stats.md 2>&1; tail -1 $O/train_kernel_stats.md
timeout 300 python bench.py --mode train --steps 20 --warmup 5 > $O/train.json 2> $O/train.err; cut -c1-120 $O/train.json
find $O -name "*.db" -delete
bash tools/pmc_march.sh f16f6 r6m/pmc > $O/pmc_out.txt 2>&1
bash tools/pmc_traffic.sh f16f6 r6m/traffic > $O/traffic_out.txt 2>&1
python bench.py --steps 1500 --warmup 3 --no-cpu-baseline --no-extras > $O/loop.log 2>&1 &
sleep 14; for i in 1 2 3; do rocm-smi --showpower --showclocks >> $O/smi_load.txt 2>&1; sleep 2; done; wait
grep -i "sclk\|power (W)" $O/smi_load.txt | head -8; tail -1 $O/loop.log | cut -c1-200
timeout 900 python bench.py --mode fullview-parity > $O/fullview_parity.json 2> $O/fullview_parity.err; cut -c1-200 $O/fullview_parity.json
timeout 600 python bench.py --mode fullview-parity --size 1024 --samples 128 --n-check 16384 > $O/fullview_parity_1024x128.json 2> $O/fp1024.err; cut -c1-200 $O/fullview_parity_1024x128.json
timeout 600 python bench.py --size 1024 --samples 128 --no-extras > $O/bench_1024x128.json 2> $O/b1024.err; cut -c1-200 $O/bench_1024x128.json
timeout 900 python bench.py --mode turntable --size 1024 --samples 128 --steps 144 --warmup 3 > $O/turntable_1024x128_144.json 2> $O/tt.err; cut -c1-200 $O/turntable_1024x128_144.json
