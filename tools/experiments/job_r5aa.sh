# round 5, job aa: the backward-input products of the STRIDED encoder layers on the matrix-pipe convolution kernels (nb_enc_conv16 with
# stride = -2: transposed gather) instead of the exact-fp32 kernel: gradient tests, training step with and without (NB_ENC_SPLIT_STRIDED=0)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5aa; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu -k "backward or train or gradients or encoder or conv" > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -3; grep "worst" $O/pytest.txt | tail -3
for v in 1 0 1 0; do
  NB_ENC_SPLIT_STRIDED=$v timeout 300 python bench.py --mode train --steps 20 --warmup 5 > $O/train_$v.json 2> $O/train_$v.err; echo "strided on pipe = $v: $(cut -c1-90 $O/train_$v.json)"
done
timeout 600 rocprofv3 --kernel-trace -d $O/tr -o t -- python bench.py --mode train --steps 10 --warmup 3 > $O/tr.log 2>&1
python tools/rocpd_summary.py $(find $O/tr -name "*.db" | head -1) > $O/train_kernel_stats.md 2>&1; tail -1 $O/train_kernel_stats.md
grep -E "conv_bwd_in|conv16" $O/train_kernel_stats.md | cut -d'|' -f2-6 | cut -c1-160
find $O -name "*.db" -delete
