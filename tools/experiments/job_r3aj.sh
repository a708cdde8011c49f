cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3aj; mkdir -p $O
(
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_parity.py -q -x -m gpu -k "backward or sgemm or train or enc or golden" 2>&1 | tail -3
timeout 300 python bench.py --mode train --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-200
rocprofv3 --kernel-trace -d $O/prof -o x -- python bench.py --mode train --steps 6 --warmup 2 > /dev/null 2>&1
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_summary.py $DB 2>/dev/null > $O/train_kernel_stats.md; sed -n 5,30p $O/train_kernel_stats.md | cut -c1-150
find gpurun_out -name "*.db" -delete
) > $O/log.txt 2>&1
cat $O/log.txt
