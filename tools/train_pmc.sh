#!/bin/bash
# PMC counters of the training step's kernels (wgrad): MFMA busy, waits, LDS, HBM fetch.  usage: bash tools/train_pmc.sh <tag>
set -e
TAG=${1:-x}
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/tpmc_${TAG}_1 -- python3 $R/bench.py --config train --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/tpmc_${TAG}_1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/tpmc_${TAG}_2 -- python3 $R/bench.py --config train --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/tpmc_${TAG}_2.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/tpmc_${TAG}_3 -- python3 $R/bench.py --config train --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/tpmc_${TAG}_3.log 2>&1
cd $R
for i in 1 2 3; do python3 tools/pmc_summary.py gpurun_out/tpmc_${TAG}_$i conv_wgrad_kernel; done > gpurun_out/tpmc_${TAG}_summary.txt
rm -rf gpurun_out/tpmc_${TAG}_1 gpurun_out/tpmc_${TAG}_2 gpurun_out/tpmc_${TAG}_3
grep -A12 "Li2ELi2ELi2ELi2ELb1E\|<2, 2, 2, 2, true>" gpurun_out/tpmc_${TAG}_summary.txt | head -60
