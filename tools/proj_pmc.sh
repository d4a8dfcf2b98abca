#!/bin/bash
# Issue / wait / LDS / MFMA counters of ffsr_tok_proj_f32's kernel (tools/proj_bench.py, one case).  Three rocprofv3 --pmc passes,
# no other trace domains; the program directly after `--`.   usage (GPU box): bash tools/proj_pmc.sh <tag> <waves> fused|plain
set -e
TAG=${1:-x}
WV=${2:-4}
CASE=${3:-plain}
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/ppmc_${TAG}_1 -- python3 $R/tools/proj_bench.py $WV $CASE > $R/gpurun_out/ppmc_${TAG}_1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/ppmc_${TAG}_2 -- python3 $R/tools/proj_bench.py $WV $CASE > $R/gpurun_out/ppmc_${TAG}_2.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_FLAT --output-format csv -d $R/gpurun_out/ppmc_${TAG}_3 -- python3 $R/tools/proj_bench.py $WV $CASE > $R/gpurun_out/ppmc_${TAG}_3.log 2>&1 || echo "pass 3 failed"
cd $R
for i in 1 2 3; do python3 tools/pmc_summary.py gpurun_out/ppmc_${TAG}_$i tok_chain_kernel; done > gpurun_out/ppmc_${TAG}_summary.txt 2>&1 || true
rm -rf gpurun_out/ppmc_${TAG}_1 gpurun_out/ppmc_${TAG}_2 gpurun_out/ppmc_${TAG}_3
cat gpurun_out/ppmc_${TAG}_summary.txt
