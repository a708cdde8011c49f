set -x
mkdir -p gpurun_out/r3a
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "f16f6" > gpurun_out/r3a/parity_f16f6.log 2>&1
tail -5 gpurun_out/r3a/parity_f16f6.log
NB_LIB_PATH=$PWD/neuralbody_amd/lib/libnb_hip_tap.so timeout 300 python tools/experiments/ms6_tap_check.py > gpurun_out/r3a/tap.log 2>&1
tail -15 gpurun_out/r3a/tap.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r3a/bench_ms6.json 2> gpurun_out/r3a/bench_ms6.err
tail -2 gpurun_out/r3a/bench_ms6.err; cat gpurun_out/r3a/bench_ms6.json | head -c 1500
timeout 300 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --precision f16f6r > gpurun_out/r3a/bench_ring.json 2> gpurun_out/r3a/bench_ring.err
cat gpurun_out/r3a/bench_ring.json | head -c 600
NB_LIB_PATH=$PWD/neuralbody_amd/lib/libnb_hip_r16.so timeout 300 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r3a/bench_ms6_r16.json 2> gpurun_out/r3a/bench_ms6_r16.err
cat gpurun_out/r3a/bench_ms6_r16.json | head -c 600
