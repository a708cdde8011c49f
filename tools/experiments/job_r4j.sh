cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r4j
timeout 900 python -m pytest tests/test_gpu_fold.py tests/test_novel_view.py tests/test_gpu_backward.py -x -q -m gpu > gpurun_out/r4j/pytest.txt 2>&1; grep -E "passed|failed|error" gpurun_out/r4j/pytest.txt | tail -3
rocprofv3 --kernel-trace --stats -d gpurun_out/r4j/stats -o s -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/r4j/stats.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/r4j/stats -name "*.db") > gpurun_out/r4j/kernel_stats.md 2>&1
python tools/rocpd_timeline.py $(find gpurun_out/r4j/stats -name "*.db") > gpurun_out/r4j/timeline.md 2>&1
find gpurun_out/r4j -name "*.db" -delete
tail -4 gpurun_out/r4j/timeline.md
timeout 300 python bench.py --mode train --steps 10 --warmup 3 2>/dev/null | cut -c1-200
