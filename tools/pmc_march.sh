#!/bin/bash
# PMC passes over one bench.py launch of the march kernel (run on the GPU box): tools/pmc_march.sh <precision> <tag>
# counters are collected with --kernel-trace only (no other trace domain), 8 SQ counters per pass
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
P=${1:-f16f6}; TAG=${2:-pmc}
rm -rf gpurun_out/${TAG}_*
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY -d gpurun_out/${TAG}_1 -o p -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --no-overlap --precision $P > gpurun_out/${TAG}_1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY -d gpurun_out/${TAG}_2 -o p -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --no-overlap --precision $P > gpurun_out/${TAG}_2.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA -d gpurun_out/${TAG}_3 -o p -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --no-overlap --precision $P > gpurun_out/${TAG}_3.log 2>&1
for d in gpurun_out/${TAG}_1 gpurun_out/${TAG}_2 gpurun_out/${TAG}_3; do python tools/pmc_print.py $(find $d -name "*.db") 2>&1 | grep -i "march" ; done > gpurun_out/${TAG}_summary.txt
cat gpurun_out/${TAG}_summary.txt
find gpurun_out -name "*.db" -delete
