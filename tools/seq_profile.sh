#!/bin/bash
# rocprofv3 kernel-trace stats of the sequential-experts pass only (quick look).  usage: bash tools/seq_profile.sh <tag>
set -e
TAG=${1:-x}
R=$PWD
cd /tmp && export TMPDIR=/tmp
FFSR_CONCURRENT_EXPERTS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_seq -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_${TAG}_seq.log 2>&1
cd $R
f=$(ls gpurun_out/prof_${TAG}_seq/*/*_kernel_stats.csv | head -1)
cp $f gpurun_out/prof_${TAG}_seq_kernel_stats.csv
rm -rf gpurun_out/prof_${TAG}_seq
python3 tools/summarize_prof.py gpurun_out/prof_${TAG}_seq_kernel_stats.csv gpurun_out/prof_${TAG}_seq.md "FFSR_CONCURRENT_EXPERTS=0 rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline (7 pipeline passes in the run)"
head -60 gpurun_out/prof_${TAG}_seq.md | cut -c1-160
