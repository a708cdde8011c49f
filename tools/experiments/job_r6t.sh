# round 6, job t: the stream a (1, 4) tiling would need on the shipped organisation (every second piece not loaded): fc_1 / fc_2 only | all phases
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6t; mkdir -p $O; rm -f $O/time.log
for v in "" _HALFSTREAM _HALFSTREAM_ALL "" _HALFSTREAM _HALFSTREAM_ALL; do
  echo "== variant '$v'" >> $O/time.log
  NB_LIB_PATH=neuralbody_amd/lib/libnb_hip$v.so NB_LAST_SAMPLE_FIXUP=0 timeout 300 python tools/experiments/fold_check.py time >> $O/time.log 2>&1
done
grep -E "variant|march" $O/time.log
