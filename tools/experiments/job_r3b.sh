cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r3b
tools/experiments/abl_ms6.sh run > gpurun_out/r3b/abl.log 2>&1
cat gpurun_out/r3b/abl.log
tools/pmc_march.sh f16f6 r3b/pmc > /dev/null 2>&1
cat gpurun_out/r3b/pmc_summary.txt
