cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r3ae
(
timeout 300 python bench.py --mode train --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-400
rocprofv3 --kernel-trace -d gpurun_out/r3ae/prof -o x -- python bench.py --mode train --steps 6 --warmup 2 > /dev/null 2>&1
DB=$(find gpurun_out/r3ae/prof -name "*.db" | head -1)
python tools/rocpd_summary.py $DB 2>/dev/null | head -60
find gpurun_out -name "*.db" -delete
) > gpurun_out/r3ae/log.txt 2>&1
cat gpurun_out/r3ae/log.txt
