cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r4n
timeout 600 python tools/experiments/points_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4n/points.txt
for rep in 1 2; do
echo "== prev"; NB_LIB_PATH=$PWD/neuralbody_amd/lib/libnb_hip_prev.so timeout 300 python tools/experiments/fold_check.py time 2>&1 | grep "^time"
echo "== new"; timeout 300 python tools/experiments/fold_check.py time 2>&1 | grep "^time"
done
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r4n/pytest.txt 2>&1; grep -E "passed|failed|error" gpurun_out/r4n/pytest.txt | tail -3
