cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r3k
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*\|SQC_[A-Z0-9_]*" | sort -u > gpurun_out/r3k/counters.txt
wc -l gpurun_out/r3k/counters.txt
grep -i "ifetch\|icache\|inst_cache\|SQC_\|INST_LEVEL\|WAVE_DEP\|IFETCH" gpurun_out/r3k/counters.txt | tr '\n' ' '
