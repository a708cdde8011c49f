# round 5, job w: the encoder after the convolutions' rework (sums reduced per workgroup, scheduled multiply, staged rows, the 8-wave
# two-group kernel gone): all GPU tests, per-launch times against the library of the previous commit (_old), serial timeline, training step
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5w; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -3
timeout 400 python tools/experiments/conv_variants.py 16:16,16:32,32:32,32:64,64:64,64:128,128:128 _old > $O/variants.log 2>&1; grep -v "Warn\|warn\|amdgpu.ids" $O/variants.log
for v in "" _old; do
  NB_LIB_PATH=neuralbody_amd/lib/libnb_hip${v}.so timeout 600 rocprofv3 --kernel-trace -d $O/tl$v -o t -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-overlap > $O/tl$v.log 2>&1
  python tools/rocpd_timeline.py $(find $O/tl$v -name "*.db" | head -1) > $O/step_timeline$v.md 2>&1; tail -1 $O/step_timeline$v.md
  NB_LIB_PATH=neuralbody_amd/lib/libnb_hip${v}.so timeout 300 python bench.py --mode train --steps 20 --warmup 5 > $O/train$v.json 2> $O/train$v.err; cut -c1-120 $O/train$v.json
done
find $O -name "*.db" -delete
