# round 5, job c: GPU tests after the lazy-volume / compact fold build / ADVICE changes, bench with the strong-scaling proxy
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5c; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-400 $O/bench.json; tail -3 $O/bench.err
