#!/bin/bash
# round 6: the whole GPU suite on the fix-up tree, every ray of a 512^2 view and 16 384 rays of a 1024^2 x 128 view against the oracle
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( time python -m pytest tests -x -q -m gpu -s ) > gpurun_out/r6b_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r6b_tests.log
python bench.py --mode fullview-parity > gpurun_out/r6b_fullview_parity.json 2> gpurun_out/r6b_fullview_parity.err
python bench.py --mode fullview-parity --size 1024 --samples 128 --n-check 16384 > gpurun_out/r6b_fullview_parity_1024x128.json 2> gpurun_out/r6b_fullview_parity_1024.err
tail -5 gpurun_out/r6b_tests.log
