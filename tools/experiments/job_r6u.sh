# round 6, job u: cache-policy bits on the weight stream's buffer loads (aux: 1 = sc0, 2 = nt, 16 = sc1 and sums), march time A/B + parity
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6u; mkdir -p $O; rm -f $O/time.log
for v in "" _AUX1 _AUX2 _AUX3 _AUX16 _AUX17 _AUX18 ""; do
  echo "== variant '$v'" >> $O/time.log
  NB_LIB_PATH=neuralbody_amd/lib/libnb_hip$v.so NB_LAST_SAMPLE_FIXUP=0 timeout 300 python tools/experiments/fold_check.py time >> $O/time.log 2>&1
done
grep -E "variant|march" $O/time.log
