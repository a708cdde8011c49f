cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r4f
for stage in points small full time; do
  timeout 300 python tools/experiments/fold_check.py $stage 2>&1 | grep -v amdgpu.ids > gpurun_out/r4f/$stage.txt
  echo "== $stage"; tail -6 gpurun_out/r4f/$stage.txt
done
NB_LIB_PATH=$PWD/neuralbody_amd/lib/libnb_hip_timing.so timeout 300 python tools/experiments/fold_phase_times.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4f/phases.txt
cat gpurun_out/r4f/phases.txt
timeout 300 python bench.py --mode turntable --steps 8 --warmup 2 2>/dev/null | cut -c1-300
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
