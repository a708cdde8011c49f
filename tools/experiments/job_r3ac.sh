cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r3ac
timeout 300 python tools/experiments/cpu_ahead.py > gpurun_out/r3ac/log.txt 2>&1
cat gpurun_out/r3ac/log.txt
