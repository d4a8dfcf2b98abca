#!/bin/bash
# Issue / LDS / MFMA counters of the fused token-chain kernel (tools/tok_bench.py at one shape).  Two rocprofv3 --pmc passes
# (8 SQ slots each), no other trace domains; the program directly after `--`.   usage (GPU box): bash tools/tok_pmc.sh <tag> [K H]
set -e
TAG=${1:-x}
K=${2:-180}
H=${3:-360}
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/tpmc_${TAG}_1 -- python3 $R/tools/tok_bench.py $K $H > $R/gpurun_out/tpmc_${TAG}_1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/tpmc_${TAG}_2 -- python3 $R/tools/tok_bench.py $K $H > $R/gpurun_out/tpmc_${TAG}_2.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $R/gpurun_out/tpmc_${TAG}_3 -- python3 $R/tools/tok_bench.py $K $H > $R/gpurun_out/tpmc_${TAG}_3.log 2>&1 || echo "pass 3 failed"
cd $R
for i in 1 2 3; do python3 tools/pmc_summary.py gpurun_out/tpmc_${TAG}_$i tok_chain_kernel; done > gpurun_out/tpmc_${TAG}_summary.txt 2>&1 || true
rm -rf gpurun_out/tpmc_${TAG}_1 gpurun_out/tpmc_${TAG}_2 gpurun_out/tpmc_${TAG}_3
cat gpurun_out/tpmc_${TAG}_summary.txt
