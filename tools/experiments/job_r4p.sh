cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r4p
timeout 600 python -m pytest tests/test_gpu_frames.py tests/test_novel_view.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r4p/pytest.txt
timeout 600 python bench.py > gpurun_out/r4p/bench.json 2> gpurun_out/r4p/bench.err; tail -c 3000 gpurun_out/r4p/bench.json
timeout 300 python bench.py --no-overlap --no-extras --no-cpu-baseline > gpurun_out/r4p/bench_serial.json 2>> gpurun_out/r4p/bench.err; cut -c1-400 gpurun_out/r4p/bench_serial.json
