# round 6, job s: the GPU suite + smoke() on the final tree (the record the driver's round-end run can be read against)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6s; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -s > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
