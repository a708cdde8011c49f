# round 5, job d: GPU tests on the tree with lazy volumes / arena / batched packs / chunked XCD remap; bench; XCD-chunk A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5d; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json; tail -2 $O/bench.err
for v in "" _xcd512; do echo "== variant '$v'" >> $O/time.log; NB_LIB_PATH=neuralbody_amd/lib/libnb_hip$v.so timeout 300 python tools/experiments/fold_check.py time >> $O/time.log 2>&1; done
grep -E "variant|march" $O/time.log
timeout 600 python bench.py --mode train --steps 20 --warmup 5 > $O/train.json 2> $O/train.err; cut -c1-200 $O/train.json
