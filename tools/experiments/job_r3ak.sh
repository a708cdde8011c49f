cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for i in 1 2; do for v in tn2 tn4 "" tn16; do so=neuralbody_amd/lib/libnb_hip${v:+_$v}.so; echo "$so: $(NB_LIB_PATH=$PWD/$so timeout 300 python bench.py --mode train --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-75)"; done; done
