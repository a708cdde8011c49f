cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r4i
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r4i/pytest.txt 2>&1; grep -E "passed|failed|error" gpurun_out/r4i/pytest.txt | tail -3
