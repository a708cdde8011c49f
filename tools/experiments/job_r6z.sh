# round 6, job z: fp4 block scale chosen by squared error (fits-the-maximum or one binade lower): parity of the bench view, the small fixture, the zoomed camera
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6z; mkdir -p $O; rm -f $O/parity.log
for v in "" _SCALEOPT; do
  echo "== variant '$v'" >> $O/parity.log
  NB_LIB_PATH=neuralbody_amd/lib/libnb_hip$v.so timeout 600 python tools/experiments/fold_check.py full small >> $O/parity.log 2>&1
done
grep -E "variant|parity|fixture|zoomed camera f16f6" $O/parity.log | cut -c1-200
