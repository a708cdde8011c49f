cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r4k
bash tools/pmc_march.sh f16f6 r4k/pmc > gpurun_out/r4k/pmc_out.txt 2>&1
bash tools/pmc_traffic.sh f16f6 r4k/traffic > gpurun_out/r4k/traffic_out.txt 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/r4k/stats -o s -- python bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r4k/stats.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/r4k/stats -name "*.db") > gpurun_out/r4k/kernel_stats.md 2>&1
find gpurun_out/r4k -name "*.db" -delete
rocm-smi --showpower --showclocks > gpurun_out/r4k/smi_idle.txt 2>&1
python bench.py --steps 1500 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r4k/loop.log 2>&1 &
sleep 14; for i in 1 2 3; do rocm-smi --showpower --showclocks >> gpurun_out/r4k/smi_load.txt 2>&1; sleep 2; done; wait
grep -i "sclk\|power (W)" gpurun_out/r4k/smi_load.txt | head -8; tail -1 gpurun_out/r4k/loop.log | cut -c1-300
