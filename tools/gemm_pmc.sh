#!/bin/bash
# MFMA-busy and LDS counters of the conv / GEMM kernels over one full-size image.
# Two rocprofv3 --pmc passes (8 SQ slots each), no other trace domains; the program directly after `--`.
# usage (GPU box): bash tools/gemm_pmc.sh <tag>
set -e
TAG=${1:-x}
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/gpmc_${TAG}_1 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/gpmc_${TAG}_1.log 2>&1
echo "pass 1 done"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/gpmc_${TAG}_2 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/gpmc_${TAG}_2.log 2>&1
echo "pass 2 done"
cd $R
for k in conv_gemm_planes_kernel conv3_strip_planes_kernel conv_gemm_bf16x3_v3_kernel scan_chunk_kernel; do
  python3 tools/pmc_summary.py gpurun_out/gpmc_${TAG}_1 $k
  python3 tools/pmc_summary.py gpurun_out/gpmc_${TAG}_2 $k
done > gpurun_out/gpmc_${TAG}_summary.txt
rm -rf gpurun_out/gpmc_${TAG}_1 gpurun_out/gpmc_${TAG}_2
wc -l gpurun_out/gpmc_${TAG}_summary.txt
