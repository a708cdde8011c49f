# round 5, job b: full-view parity records, tracked records of BASELINE configs 3 and 5 (item 6), fp4 probe
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5b; mkdir -p $O
timeout 120 exp_bin/probe_fp4 > $O/probe_fp4.log 2>&1; cat $O/probe_fp4.log
timeout 900 python bench.py --mode fullview-parity > $O/fullview_512.json 2> $O/fullview_512.err; cut -c1-400 $O/fullview_512.json
timeout 600 python bench.py --mode fullview-parity --size 1024 --samples 128 --n-check 16384 > $O/fullview_1024.json 2> $O/fullview_1024.err; cut -c1-400 $O/fullview_1024.json
timeout 600 python bench.py --size 1024 --samples 128 --no-extras > $O/bench_1024x128.json 2> $O/bench_1024x128.err; cut -c1-300 $O/bench_1024x128.json
timeout 600 python bench.py --mode turntable --size 1024 --samples 128 --steps 144 --warmup 3 > $O/turntable_1024x128_144.json 2> $O/turntable.err; cut -c1-600 $O/turntable_1024x128_144.json
