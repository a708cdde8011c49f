# round 5, job v: BatchNorm sums of a workgroup's waves reduced in LDS before the fp64 atomics (all five convolution kernels), the LDS-slab
# kernel's multiply phase scheduled (fragment reads one chunk ahead, products round the accumulators) and its rows staged through LDS:
# product against the library of the previous commit (_old) and against the un-staged build (_nostage)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5v; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu -k "encoder or conv or sparse or train or backward or frames or fold" > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -2
timeout 400 python tools/experiments/conv_variants.py 64:64,64:128,128:128,32:32,32:64 _old _nostage > $O/variants.log 2>&1; grep -v "Warn\|warn\|amdgpu.ids" $O/variants.log
for rep in 1 2; do
  for v in "" _old _nostage; do
    echo "== variant '${v}' rep $rep" >> $O/ab.log
    NB_LIB_PATH=neuralbody_amd/lib/libnb_hip${v}.so timeout 300 python tools/experiments/encoder_time.py train >> $O/ab.log 2>&1
  done
done
grep -v "Warn\|warn\|amdgpu.ids" $O/ab.log
for v in "" _old; do
  NB_LIB_PATH=neuralbody_amd/lib/libnb_hip${v}.so timeout 600 rocprofv3 --kernel-trace -d $O/tl$v -o t -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-overlap > $O/tl$v.log 2>&1
  python tools/rocpd_timeline.py $(find $O/tl$v -name "*.db" | head -1) > $O/step_timeline$v.md 2>&1; tail -1 $O/step_timeline$v.md
done
find $O -name "*.db" -delete
