cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r4e
NB_LIB_PATH=$PWD/neuralbody_amd/lib/libnb_hip_timing.so timeout 300 python tools/experiments/fold_phase_times.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4e/phases.txt
cat gpurun_out/r4e/phases.txt
rocprofv3 --kernel-trace --stats -d gpurun_out/r4e/tt -o s -- python bench.py --mode turntable --steps 8 --warmup 2 > gpurun_out/r4e/tt.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/r4e/tt -name "*.db") > gpurun_out/r4e/tt_stats.md 2>&1
find gpurun_out/r4e -name "*.db" -delete
head -12 gpurun_out/r4e/tt_stats.md | cut -c1-200; tail -1 gpurun_out/r4e/tt.log | cut -c1-400
