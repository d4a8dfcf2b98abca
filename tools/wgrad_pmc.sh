#!/bin/bash
# PMC counters of tools/wgrad_bench.py's kernels (f32 MFMA vs transposing-read bf16 weight gradient).  usage: bash tools/wgrad_pmc.sh <tag>
set -e
TAG=${1:-x}
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/wpmc_${TAG}_1 -- python3 $R/tools/wgrad_bench.py > $R/gpurun_out/wpmc_${TAG}_1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/wpmc_${TAG}_2 -- python3 $R/tools/wgrad_bench.py > $R/gpurun_out/wpmc_${TAG}_2.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/wpmc_${TAG}_3 -- python3 $R/tools/wgrad_bench.py > $R/gpurun_out/wpmc_${TAG}_3.log 2>&1
cd $R
for i in 1 2 3; do python3 tools/pmc_summary.py gpurun_out/wpmc_${TAG}_$i conv_wgrad; done > gpurun_out/wpmc_${TAG}_summary.txt
rm -rf gpurun_out/wpmc_${TAG}_1 gpurun_out/wpmc_${TAG}_2 gpurun_out/wpmc_${TAG}_3
cat gpurun_out/wpmc_${TAG}_summary.txt
