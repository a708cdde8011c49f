# round 5, job ab: the training step's records on the final tree (kernel trace summary + the bench line), with the clock under load
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5ab; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace -d $O/tr -o t -- python bench.py --mode train --steps 10 --warmup 3 > $O/tr.log 2>&1
python tools/rocpd_summary.py $(find $O/tr -name "*.db" | head -1) > $O/train_kernel_stats.md 2>&1; tail -1 $O/train_kernel_stats.md
find $O -name "*.db" -delete
python bench.py --mode train --steps 400 --warmup 5 > $O/train_loop.json 2> $O/train_loop.err &
sleep 9; rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -2; wait
timeout 300 python bench.py --mode train --steps 20 --warmup 5 > $O/train.json 2> $O/train.err; cut -c1-120 $O/train.json
