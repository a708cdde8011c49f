# round 5, job z: row stages per wave in the LDS-slab convolution — as many as fit (product: 2 for 64-channel rows, 1 for 128-channel
# ones), forced to 1 (_d1) and to 0 (_d0): tests, then the kernels' averages in the pipeline (rocprofv3 --kernel-trace --stats, 8 steps)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5z; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu -k "encoder or conv or sparse or train or backward" > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -2
for v in "" _d0 _d1 ""; do
  NB_LIB_PATH=neuralbody_amd/lib/libnb_hip${v}.so timeout 600 rocprofv3 --kernel-trace --stats -d $O/st$v -o s -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-overlap > $O/st$v.log 2>&1
  python tools/rocpd_summary.py $(find $O/st$v -name "*.db" | head -1) > $O/kernel_stats$v.md 2>&1
  echo "== variant '${v}'"; grep -E "conv16_lds" $O/kernel_stats$v.md | cut -d'|' -f2-6 | cut -c1-150
  find $O -name "*.db" -delete
done
