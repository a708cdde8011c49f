#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( time python -m pytest tests/test_gpu_frames.py tests/test_gpu_fold.py tests/test_gpu_parity.py -q -m gpu -s -k "batch_of_two or later_frame or edge_cases" ) > gpurun_out/r6d_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r6d_tests.log
grep -n "passed\|failed\|batch of two" gpurun_out/r6d_tests.log | tail -5
