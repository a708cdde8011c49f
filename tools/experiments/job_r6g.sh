# round 6, job g: fp4 cross-term weights: none (round 5's stream) | W_h only | both, A/B on one box: march time + parity of the bench view
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6g; mkdir -p $O; rm -f $O/*.log
for v in _F0 _F1 "" _F0 _F1 ""; do
  echo "== variant '$v'" >> $O/time.log
  NB_LIB_PATH=neuralbody_amd/lib/libnb_hip$v.so NB_LAST_SAMPLE_FIXUP=0 timeout 300 python tools/experiments/fold_check.py time >> $O/time.log 2>&1
done
for v in _F0 _F1 ""; do
  echo "== variant '$v'" >> $O/parity.log
  NB_LIB_PATH=neuralbody_amd/lib/libnb_hip$v.so timeout 600 python tools/experiments/fold_check.py full small >> $O/parity.log 2>&1
done
grep -E "variant|march" $O/time.log; grep -vi "warn" $O/parity.log | tail -30
