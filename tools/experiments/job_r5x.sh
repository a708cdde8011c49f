# round 5, job x: the LDS-slab convolution IN THE PIPELINE (inputs fresh from the BatchNorm kernel, not hot in L2 as in conv_variants.py):
# rows straight into registers (_f0), staged with the fetches in front of the MFMAs (_f1), staged with the fetches between the MFMAs
# (product), the previous commit (_old); averages over 8 steps from the kernel trace
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5x; mkdir -p $O
for v in "" _f0 _f1 _old "" _f1; do
  NB_LIB_PATH=neuralbody_amd/lib/libnb_hip${v}.so timeout 600 rocprofv3 --kernel-trace --stats -d $O/st$v -o s -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-overlap > $O/st$v.log 2>&1
  python tools/rocpd_summary.py $(find $O/st$v -name "*.db" | head -1) > $O/kernel_stats$v.md 2>&1
  echo "== variant '${v}'"; grep -E "conv16_lds|conv16_kernel|conv16_ks|conv_kernel" $O/kernel_stats$v.md | cut -d'|' -f2-6 | cut -c1-150
  find $O -name "*.db" -delete
done
