# round 6, job k: the activation tap written in whole rows: backward / stage tests, training step time and kernel stats
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6k; mkdir -p $O
python -m pytest tests/test_gpu_backward.py tests/test_gpu_parity.py -q -m gpu -k "backward or stages or training or gradients or decode" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
python bench.py --mode train --steps 20 --warmup 5 > $O/train.json 2> $O/train.err
timeout 600 rocprofv3 --kernel-trace -d $O/tr -o t -- python bench.py --mode train --steps 10 --warmup 3 > $O/tr.log 2>&1
python tools/rocpd_summary.py $(find $O/tr -name "*.db" | head -1) > $O/train_kernel_stats.md 2>&1
find $O -name "*.db" -delete
grep -E "passed|failed|rc=" $O/tests.log | tail -3; cut -c1-160 $O/train.json; grep -E "points_kernel|^\| kernel|total" $O/train_kernel_stats.md | head -5; tail -2 $O/train_kernel_stats.md
