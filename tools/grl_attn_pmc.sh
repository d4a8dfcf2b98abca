#!/bin/bash
# Vector-ALU / LDS / memory counters of GRL's attention kernels.  Two rocprofv3 --pmc passes, no other trace
# domains; the program directly after `--`.   usage (GPU box): bash tools/grl_attn_pmc.sh <tag>
set -e
TAG=${1:-x}
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/gattn_${TAG}_1 -- python3 $R/tools/grl_attn_bench.py > $R/gpurun_out/gattn_${TAG}_1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/gattn_${TAG}_2 -- python3 $R/tools/grl_attn_bench.py > $R/gpurun_out/gattn_${TAG}_2.log 2>&1
cd $R
for i in 1 2; do python3 tools/pmc_summary.py gpurun_out/gattn_${TAG}_$i grl_; done > gpurun_out/gattn_${TAG}_summary.txt 2>&1
rm -rf gpurun_out/gattn_${TAG}_1 gpurun_out/gattn_${TAG}_2
cat gpurun_out/gattn_${TAG}_1.log gpurun_out/gattn_${TAG}_summary.txt
