cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r4m
timeout 600 python tools/experiments/points_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4m/points.txt
