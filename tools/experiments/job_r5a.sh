# round 5, job a: premises of the march restructure, A/B on one box (march time of the 512 x 512 x 64 bench view)
#   base | weight pieces all from the same 8 KiB (no L2 -> L1 stream) | no weight loads at all | one workgroup per CU at 256
#   registers | one workgroup per CU at 512 registers
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5a; mkdir -p $O
for v in "" _SAMEADDR _NOLOAD _ONEWG _ONEWG512 ""; do
  echo "== variant '$v'" >> $O/time.log
  NB_LIB_PATH=neuralbody_amd/lib/libnb_hip$v.so timeout 300 python tools/experiments/fold_check.py time >> $O/time.log 2>&1
done
grep -E "variant|march" $O/time.log
