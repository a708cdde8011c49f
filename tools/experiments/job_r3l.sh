cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r3l
tools/pmc_icache.sh f16f6 r3l/ms6 > /dev/null 2>&1; cat gpurun_out/r3l/ms6_summary.txt | cut -c1-140
tools/pmc_icache.sh f16f6r r3l/ring > /dev/null 2>&1; cat gpurun_out/r3l/ring_summary.txt | cut -c1-140
