cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r4a
for stage in rows points small full time; do
  timeout 300 python tools/experiments/fold_check.py $stage 2>&1 | grep -v amdgpu.ids > gpurun_out/r4a/$stage.txt
  echo "== $stage rc=$?"; tail -25 gpurun_out/r4a/$stage.txt
done
NB_LIB_PATH=$PWD/neuralbody_amd/lib/libnb_hip_tap.so timeout 300 python tools/experiments/fold_check.py tap 2>&1 | grep -v amdgpu.ids > gpurun_out/r4a/tap.txt
echo "== tap"; tail -30 gpurun_out/r4a/tap.txt
