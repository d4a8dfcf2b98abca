#!/bin/bash
# PMC counters of the scan kernels (two passes).  Run on the GPU box: bash tools/scan_pmc.sh
set -e
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/scan_pmc1 -- python3 $R/tools/scan_bench.py 352 > $R/gpurun_out/scan_pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/scan_pmc2 -- python3 $R/tools/scan_bench.py 352 > $R/gpurun_out/scan_pmc2.log 2>&1
cd $R
python3 tools/pmc_summary.py gpurun_out/scan_pmc1 scan_chunk > gpurun_out/scan_pmc_summary.txt
python3 tools/pmc_summary.py gpurun_out/scan_pmc2 scan_chunk >> gpurun_out/scan_pmc_summary.txt
