# round 6, job q: calibrate rocprofv3's WRITE_SIZE on the march's store patterns (tools/experiments/probe_write.hip)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6q; mkdir -p $O exp_bin
hipcc -O3 --offload-arch=gfx950 tools/experiments/probe_write.hip -o exp_bin/probe_write 2> $O/build.log || { cat $O/build.log; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/w -o w -- exp_bin/probe_write > $O/run.log 2>&1
python tools/pmc_print.py $(find $O/w -name "*.db" | head -1) > $O/write_size.txt 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $O/w2 -o w -- exp_bin/probe_write > $O/run2.log 2>&1
python tools/pmc_print.py $(find $O/w2 -name "*.db" | head -1) > $O/wrreq.txt 2>&1
find $O -name "*.db" -delete
cat $O/write_size.txt; cat $O/wrreq.txt | head -20; tail -2 $O/run.log
# ... and WRITE_SIZE of the SPILL-FREE instance of the same kernel (nb_decode_points on 2.1 M points: MODE 1, 0 spilled registers, 33.5 MB
# of raw [n, 4] written as one 16-byte store per point) against the ray march's (25 spilled registers, 80 B of scratch per lane)
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/p -o w -- python tools/experiments/points_bench.py > $O/points.log 2>&1
python tools/pmc_print.py $(find $O/p -name "*.db" | head -1) 2>&1 | grep -i "march_fold\|points" > $O/points_write_size.txt
find $O -name "*.db" -delete
cat $O/points_write_size.txt
