cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4u; mkdir -p $O
timeout 300 python tools/experiments/overlap_check.py --trace 2>&1 | grep -v amdgpu.ids | tee $O/trace.txt
for i in 1 2 3; do timeout 300 python bench.py --mode train 2>/dev/null | cut -c1-120; done | tee $O/train.txt
