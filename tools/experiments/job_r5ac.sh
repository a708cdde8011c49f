# round 5, job ac: the strided levels' index sets in three launches (nb_enc_downsample_index_all, ABI 19) instead of three per level:
# all GPU tests, then the serial timeline with it and without (NB_INDEX_ALL=0)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5ac; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -3
for v in 1 0; do
  NB_INDEX_ALL=$v timeout 600 rocprofv3 --kernel-trace -d $O/tl$v -o t -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-overlap > $O/tl$v.log 2>&1
  python tools/rocpd_timeline.py $(find $O/tl$v -name "*.db" | head -1) > $O/step_timeline$v.md 2>&1; echo "NB_INDEX_ALL=$v: $(tail -1 $O/step_timeline$v.md)"
done
find $O -name "*.db" -delete
timeout 300 python bench.py --mode train --steps 20 --warmup 5 > $O/train.json 2> $O/train.err; cut -c1-100 $O/train.json
