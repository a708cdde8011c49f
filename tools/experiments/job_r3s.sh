cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r3s
rocprofv3 --kernel-trace -d gpurun_out/r3s/prof -o x -- python bench.py --steps 4 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r3s/bench.log 2>&1
DB=$(find gpurun_out/r3s/prof -name "*.db" | head -1)
python tools/rocpd_timeline.py $DB nb_march > gpurun_out/r3s/step_timeline.md
tail -3 gpurun_out/r3s/step_timeline.md
find gpurun_out -name "*.db" -delete
