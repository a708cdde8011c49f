# round 5, job q: __launch_bounds__(256, 2) on the encoder's and the backward's 256-thread kernels (hipcc took up to 512 registers
# where nothing asked for two waves per SIMD): GPU tests, encoder timeline, training step + kernel stats
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5q; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -2
timeout 600 rocprofv3 --kernel-trace -d $O/tl -o t -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-overlap > $O/tl.log 2>&1
python tools/rocpd_timeline.py $(find $O/tl -name "*.db" | head -1) > $O/step_timeline.md 2>&1; tail -2 $O/step_timeline.md
timeout 600 rocprofv3 --kernel-trace -d $O/tr -o t -- python bench.py --mode train --steps 10 --warmup 3 > $O/tr.log 2>&1
python tools/rocpd_summary.py $(find $O/tr -name "*.db" | head -1) > $O/train_kernel_stats.md 2>&1; tail -1 $O/train_kernel_stats.md
find $O -name "*.db" -delete
timeout 300 python bench.py --mode train --steps 30 --warmup 5 > $O/train.json 2> $O/train.err; cut -c1-100 $O/train.json
