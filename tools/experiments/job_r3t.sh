cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r3t
timeout 120 exp_bin/probe_coexec2 > gpurun_out/r3t/coexec2.log 2>&1
cat gpurun_out/r3t/coexec2.log
