#!/bin/bash
# round 6, first light of the last-sample fix-up: its tests, the full-size parity tests, the default bench
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests/test_gpu_fixup.py tests/test_gpu_fullsize.py -x -q -m gpu -s -k "fixup or headline or config3" > gpurun_out/r6a_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r6a_tests.log
python bench.py > gpurun_out/r6a_bench.json 2> gpurun_out/r6a_bench.err
NB_LAST_SAMPLE_FIXUP=0 python bench.py > gpurun_out/r6a_bench_nofix.json 2> gpurun_out/r6a_bench_nofix.err
tail -5 gpurun_out/r6a_tests.log
