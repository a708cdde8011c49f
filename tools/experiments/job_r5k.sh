# round 5, job k: chunk size of the XCD remap (16 / 32 / 64 / 128); training step with the fatter BatchNorm-backward reduce
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5k; mkdir -p $O
for v in "" _xc16 _xc32 _xc128 ""; do echo "== variant '$v'" >> $O/xcd.log; NB_LIB_PATH=neuralbody_amd/lib/libnb_hip$v.so timeout 300 python tools/experiments/xcd_chunk_check.py >> $O/xcd.log 2>&1; done
grep -E "variant|full" $O/xcd.log
timeout 600 python -m pytest tests/test_gpu_backward.py -x -q > $O/pytest_bwd.txt 2>&1; tail -2 $O/pytest_bwd.txt
timeout 300 python bench.py --mode train --steps 30 --warmup 5 > $O/train.json 2> $O/train.err; cut -c1-100 $O/train.json
timeout 600 rocprofv3 --kernel-trace -d $O/tr -o t -- python bench.py --mode train --steps 10 --warmup 3 > $O/tr.log 2>&1
python tools/rocpd_summary.py $(find $O/tr -name "*.db" | head -1) > $O/train_kernel_stats.md 2>&1; grep -E "bn_bwd|Total" $O/train_kernel_stats.md | cut -c1-120
find $O -name "*.db" -delete
