cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r3v
L=$PWD/neuralbody_amd/lib
(
echo "=== paired (lag 7), 8 waves / CU"
NB_MS6_PAIR=1 NB_LIB_PATH=$L/libnb_hip_ms6TIMING.so timeout 200 python tools/experiments/ms6_phase_times.py 2>&1 | grep -v amdgpu.ids
echo "=== two independent workgroups / CU"
NB_MS6_PAIR=0 NB_LIB_PATH=$L/libnb_hip_ms6TIMING.so timeout 200 python tools/experiments/ms6_phase_times.py 2>&1 | grep -v amdgpu.ids
echo "=== one workgroup / CU"
NB_MS6_PAIR=0 NB_LIB_PATH=$L/libnb_hip_ms6TIMING_ONEWG.so timeout 200 python tools/experiments/ms6_phase_times.py 2>&1 | grep -v amdgpu.ids
) > gpurun_out/r3v/log.txt 2>&1
cat gpurun_out/r3v/log.txt
