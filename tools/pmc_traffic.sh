#!/bin/bash
# HBM traffic of the march kernel (run on the GPU box): tools/pmc_traffic.sh <precision> <tag>
# FETCH_SIZE / WRITE_SIZE / L2 hits in separate --pmc passes, --kernel-trace only (MI355X_MICROARCH.md, HBM / rocprofv3 section)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
P=${1:-f16f6}; TAG=${2:-traffic}
rm -rf gpurun_out/${TAG}_*
i=0
for ctr in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i + 1))
  rocprofv3 --kernel-trace --pmc $ctr -d gpurun_out/${TAG}_$i -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-overlap --precision $P > gpurun_out/${TAG}_$i.log 2>&1
  python tools/pmc_print.py $(find gpurun_out/${TAG}_$i -name "*.db") 2>&1 | grep -i "march"
done > gpurun_out/${TAG}_summary.txt
cat gpurun_out/${TAG}_summary.txt
find gpurun_out -name "*.db" -delete
