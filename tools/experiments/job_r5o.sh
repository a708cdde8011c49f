# round 5, job o: encoder kernels — BatchNorm + ReLU with the per-channel affine once per block, the fp32 convolution with its
# neighbour indices and rows prefetched, fold planes with 16-byte weight loads: GPU tests, timeline, bench
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5o; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -2
timeout 600 rocprofv3 --kernel-trace -d $O/tl -o t -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-overlap > $O/tl.log 2>&1
python tools/rocpd_timeline.py $(find $O/tl -name "*.db" | head -1) > $O/step_timeline.md 2>&1; tail -2 $O/step_timeline.md
find $O -name "*.db" -delete
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
timeout 300 python bench.py --mode train --steps 30 --warmup 5 > $O/train.json 2> $O/train.err; cut -c1-100 $O/train.json
