cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r4b
bash tools/pmc_march.sh f16f6v r4b/pmc > gpurun_out/r4b/pmc_out.txt 2>&1
bash tools/pmc_traffic.sh f16f6v r4b/traffic > gpurun_out/r4b/traffic_out.txt 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/r4b/stats -o s -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --precision f16f6v > gpurun_out/r4b/stats.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/r4b/stats -name "*.db") > gpurun_out/r4b/kernel_stats.md 2>&1
find gpurun_out/r4b -name "*.db" -delete
(rocm-smi --showpower --showclocks > gpurun_out/r4b/smi_idle.txt 2>&1)
python bench.py --steps 300 --warmup 3 --no-cpu-baseline --no-extras --precision f16f6v > gpurun_out/r4b/loop.log 2>&1 &
sleep 25; rocm-smi --showpower --showclocks > gpurun_out/r4b/smi_load.txt 2>&1; wait
tail -3 gpurun_out/r4b/pmc_out.txt; cat gpurun_out/r4b/traffic_out.txt | tail -5; head -30 gpurun_out/r4b/kernel_stats.md; grep -i "sclk\|power" gpurun_out/r4b/smi_load.txt | head; tail -1 gpurun_out/r4b/loop.log | cut -c1-600
