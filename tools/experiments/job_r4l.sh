cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r4l
for stage in points small full; do
  timeout 300 python tools/experiments/fold_check.py $stage 2>&1 | grep -v amdgpu.ids > gpurun_out/r4l/$stage.txt
  echo "== $stage"; tail -3 gpurun_out/r4l/$stage.txt
done
for rep in 1 2; do
echo "== prev"; NB_LIB_PATH=$PWD/neuralbody_amd/lib/libnb_hip_prev.so timeout 300 python tools/experiments/fold_check.py time 2>&1 | grep "^time"
echo "== new"; timeout 300 python tools/experiments/fold_check.py time 2>&1 | grep "^time"
done
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r4l/pytest.txt 2>&1; grep -E "passed|failed|error" gpurun_out/r4l/pytest.txt | tail -3
