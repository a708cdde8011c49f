cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r4o
timeout 300 python tools/experiments/overlap_check.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4o/overlap.txt
