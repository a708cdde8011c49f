# round 6, job e: premises of the march's remaining arithmetic economies, A/B on one box (march time of the 512 x 512 x 64 bench view)
#   base | second piece of every W_h six-bit fragment not loaded (fp4 W_h proxy) | ... of both cross fragments (fp4 both) |
#   colour head's 128 x 256 part with one cross term | no cross terms at all | publishes without operand conversion | no weight loads
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6e; mkdir -p $O; rm -f $O/time.log
for v in "" _FP4H _FP4BOTH _VG1 _NOCROSS _NOPUB _NOLOAD ""; do
  echo "== variant '$v'" >> $O/time.log
  NB_LIB_PATH=neuralbody_amd/lib/libnb_hip$v.so NB_LAST_SAMPLE_FIXUP=0 timeout 300 python tools/experiments/fold_check.py time >> $O/time.log 2>&1
done
grep -E "variant|march" $O/time.log
