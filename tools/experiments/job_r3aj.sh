cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3aj; mkdir -p $O
(
timeout 300 python bench.py --mode train --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-200
rocprofv3 --kernel-trace -d $O/prof -o x -- python bench.py --mode train --steps 6 --warmup 2 > /dev/null 2>&1
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_summary.py $DB 2>/dev/null > $O/train_kernel_stats.md; sed -n 5,34p $O/train_kernel_stats.md | cut -c1-130
find gpurun_out -name "*.db" -delete
) > $O/log.txt 2>&1
cat $O/log.txt
