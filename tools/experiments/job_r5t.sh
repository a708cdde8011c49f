# round 5, job t: the LDS-slab convolution's gathered rows staged through LDS (RowStage: 64 contiguous bytes per lane quad by LDS-DMA,
# fragments by ds_read_b128) against rows loaded straight into the fragment registers (-DNB_CONV_STAGE=0); and the 4-wave staged kernel
# in place of the 8-wave two-group one on the 128-channel levels (-DNB_CONV_LDS2=0)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5t; mkdir -p $O
for v in "" _nolds2; do
  NB_LIB_PATH=neuralbody_amd/lib/libnb_hip${v}.so timeout 900 python -m pytest tests -x -q -m gpu -k "encoder or conv or sparse or train or backward" > $O/pytest$v.txt 2>&1; grep -E "passed|failed|error" $O/pytest$v.txt | tail -2
done
for rep in 1 2; do
  for v in "" _nostage _nolds2; do
    echo "== variant '${v}' rep $rep" >> $O/ab.log
    NB_LIB_PATH=neuralbody_amd/lib/libnb_hip${v}.so timeout 300 python tools/experiments/encoder_time.py train >> $O/ab.log 2>&1
  done
done
grep -v "Warn\|warn\|amdgpu.ids" $O/ab.log
for v in "" _nostage _nolds2; do
  NB_LIB_PATH=neuralbody_amd/lib/libnb_hip${v}.so timeout 600 rocprofv3 --kernel-trace -d $O/tl$v -o t -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-overlap > $O/tl$v.log 2>&1
  python tools/rocpd_timeline.py $(find $O/tl$v -name "*.db" | head -1) > $O/step_timeline$v.md 2>&1; tail -1 $O/step_timeline$v.md
done
find $O -name "*.db" -delete
