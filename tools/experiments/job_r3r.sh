cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r3r
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "decode or density or end_to_end or trained or noise" > gpurun_out/r3r/tests.log 2>&1
tail -4 gpurun_out/r3r/tests.log; grep -n "FAILED\|Error" gpurun_out/r3r/tests.log | head
python tools/experiments/points_bench.py > gpurun_out/r3r/points_bench.log 2>&1; cat gpurun_out/r3r/points_bench.log | tail -8
