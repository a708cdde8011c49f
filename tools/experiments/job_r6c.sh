#!/bin/bash
# round 6: the whole GPU suite (fix-up, launcher, bench-scene reference fixtures, later-frame saturation)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( time python -m pytest tests -q -m gpu -s ) > gpurun_out/r6c_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r6c_tests.log
grep -n "passed\|failed" gpurun_out/r6c_tests.log | tail -3
