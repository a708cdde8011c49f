cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r4c
for stage in points small full time; do
  timeout 300 python tools/experiments/fold_check.py $stage 2>&1 | grep -v amdgpu.ids > gpurun_out/r4c/$stage.txt
  echo "== $stage"; tail -8 gpurun_out/r4c/$stage.txt
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "f16f6v" 2>&1 | tail -15 > gpurun_out/r4c/pytest.txt
cat gpurun_out/r4c/pytest.txt
