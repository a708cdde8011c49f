cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_dist.py -q -x -m gpu 2>&1 | tail -3
for i in 1 2; do timeout 300 python bench.py --mode train --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-75; done
