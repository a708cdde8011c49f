# round 5, job s: the encoder's row groups dealt to the XCDs in contiguous eighths (nb_rowgroups.h) against blockIdx.x as the row group
# (library built with -DNB_XCD_ROWS=0): encoder alone + training step, twice each, then the per-kernel durations of both
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5s; mkdir -p $O
for rep in 1 2; do
  for v in "" _noxcd; do
    echo "== variant '${v}' rep $rep" >> $O/ab.log
    NB_LIB_PATH=neuralbody_amd/lib/libnb_hip${v}.so timeout 300 python tools/experiments/encoder_time.py train >> $O/ab.log 2>&1
  done
done
cat $O/ab.log | grep -v Warn
for v in "" _noxcd; do
  NB_LIB_PATH=neuralbody_amd/lib/libnb_hip${v}.so timeout 600 rocprofv3 --kernel-trace -d $O/tl$v -o t -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-overlap > $O/tl$v.log 2>&1
  python tools/rocpd_timeline.py $(find $O/tl$v -name "*.db" | head -1) > $O/step_timeline$v.md 2>&1; tail -1 $O/step_timeline$v.md
done
find $O -name "*.db" -delete
timeout 1200 python -m pytest tests -x -q -m gpu -k "encoder or conv or train or backward or sparse" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
