# round 6, job v: the first block's scale dword of every phase requested at the top of the depth step: march time A/B + parity
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6v; mkdir -p $O; rm -f $O/time.log
for v in "" _SCEARLY "" _SCEARLY; do
  echo "== variant '$v'" >> $O/time.log
  NB_LIB_PATH=neuralbody_amd/lib/libnb_hip$v.so NB_LAST_SAMPLE_FIXUP=0 timeout 300 python tools/experiments/fold_check.py time >> $O/time.log 2>&1
done
NB_LIB_PATH=neuralbody_amd/lib/libnb_hip_SCEARLY.so timeout 600 python tools/experiments/fold_check.py full small > $O/parity.log 2>&1
grep -E "variant|march" $O/time.log; grep -E "parity|fixture|OK" $O/parity.log | cut -c1-160
