cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r4g
NB_LIB_PATH=$PWD/neuralbody_amd/lib/libnb_hip_timing.so timeout 300 python tools/experiments/fold_phase_times.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4g/phases.txt
tail -6 gpurun_out/r4g/phases.txt
