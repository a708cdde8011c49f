# round 5, job n: weights output as whole 64-byte segments (ds_bpermute regrouping): time A/B, parity, WRITE_SIZE; GPU tests after the
# FeatureVolumes change
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5n; mkdir -p $O
for v in _base "" _base ""; do echo "== variant '$v'" >> $O/time.log; NB_LIB_PATH=neuralbody_amd/lib/libnb_hip$v.so timeout 300 python tools/experiments/fold_check.py time >> $O/time.log 2>&1; done
grep -E "variant|march" $O/time.log
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -2
bash tools/pmc_traffic.sh f16f6 r5n/traffic > $O/traffic_out.txt 2>&1; cat gpurun_out/r5n/traffic_summary.txt
