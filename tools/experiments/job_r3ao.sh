cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r3ao
MS6_SUB=B NB_LIB_PATH=$PWD/neuralbody_amd/lib/libnb_hip_ms6TIMING_2.so timeout 200 python tools/experiments/ms6_phase_times.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r3ao/log.txt
cat gpurun_out/r3ao/log.txt | tail -8
