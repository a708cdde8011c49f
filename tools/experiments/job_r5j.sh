# round 5, job j: phase stamps of the 128-sample organisation
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5j; mkdir -p $O
NB_MARCH_NU=2 NB_LIB_PATH=neuralbody_amd/lib/libnb_hip_timing.so timeout 300 python tools/experiments/fold_phase_times.py > $O/phases_nu2.md 2>&1; cat $O/phases_nu2.md
NB_LIB_PATH=neuralbody_amd/lib/libnb_hip_timing.so timeout 300 python tools/experiments/fold_phase_times.py > $O/phases_nu1.md 2>&1; tail -8 $O/phases_nu1.md
