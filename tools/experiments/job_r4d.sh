cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r4d
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r4d/pytest.txt
cat gpurun_out/r4d/pytest.txt
timeout 600 python bench.py > gpurun_out/r4d/bench.json 2> gpurun_out/r4d/bench.err
tail -3 gpurun_out/r4d/bench.err; cut -c1-1500 gpurun_out/r4d/bench.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
