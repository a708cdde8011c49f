cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r3q
timeout 1200 python -m pytest tests -q -m gpu > gpurun_out/r3q/gpu_tests.log 2>&1
tail -6 gpurun_out/r3q/gpu_tests.log
grep -h "trained/\|ill-conditioned\|rgb L-inf of" gpurun_out/r3q/gpu_tests.log | head -20
grep -h "passed\|failed\|trained/auto" gpurun_out/r3q/gpu_tests.log | tail -5; grep -n "FAILED" gpurun_out/r3q/gpu_tests.log | head
