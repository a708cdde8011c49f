#!/bin/bash
# instruction-fetch / co-execution counters of the march kernel (run on the GPU box): tools/pmc_icache.sh <precision> <tag>
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
P=${1:-f16f6}; TAG=${2:-pmcic}
rm -rf gpurun_out/${TAG}_*
rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d gpurun_out/${TAG}_1 -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --precision $P > gpurun_out/${TAG}_1.log 2>&1
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQC_TC_STALL SQC_ICACHE_BUSY_CYCLES SQ_INSTS_SALU -d gpurun_out/${TAG}_2 -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --precision $P > gpurun_out/${TAG}_2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT64 -d gpurun_out/${TAG}_3 -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --precision $P > gpurun_out/${TAG}_3.log 2>&1
for d in gpurun_out/${TAG}_1 gpurun_out/${TAG}_2 gpurun_out/${TAG}_3; do python tools/pmc_print.py $(find $d -name "*.db") 2>&1 | grep -i "march" ; done > gpurun_out/${TAG}_summary.txt
cat gpurun_out/${TAG}_summary.txt
find gpurun_out -name "*.db" -delete
