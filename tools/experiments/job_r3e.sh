cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r3e
NB_LIB_PATH=$PWD/neuralbody_amd/lib/libnb_hip_ms6TIMING.so python tools/experiments/ms6_phase_times.py > gpurun_out/r3e/phases_2wg.log 2>&1
grep -v "^|" gpurun_out/r3e/phases_2wg.log
NB_LIB_PATH=$PWD/neuralbody_amd/lib/libnb_hip_ms6TIMING_ONEWG.so python tools/experiments/ms6_phase_times.py > gpurun_out/r3e/phases_1wg.log 2>&1
cat gpurun_out/r3e/phases_1wg.log
