cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_backward.py -q -x -m gpu -s 2>&1 | grep -E "passed|failed|Error|encoder|full|rel " | tail -12
for i in 1 2; do echo "split: $(timeout 300 python bench.py --mode train --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-75)"; done
echo "fp32 bwd-in: $(NB_ENC_SPLIT=0 timeout 300 python bench.py --mode train --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-75)"
