#!/bin/bash
# rocprofv3 kernel-trace stats of the cached-feature training step (bench.py --config train).  usage: bash tools/train_profile.sh <tag>
set -e
TAG=${1:-x}
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_train -- python3 $R/bench.py --config train --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_${TAG}_train.log 2>&1
cd $R
f=$(ls gpurun_out/prof_${TAG}_train/*/*_kernel_stats.csv | head -1)
cp $f gpurun_out/prof_${TAG}_train_kernel_stats.csv
python3 tools/summarize_prof.py gpurun_out/prof_${TAG}_train_kernel_stats.csv gpurun_out/prof_${TAG}_train.md "rocprofv3 --kernel-trace --stats -- python3 bench.py --config train --steps 3 --warmup 1  (4 training steps in the run)"
rm -rf gpurun_out/prof_${TAG}_train
tail -2 gpurun_out/prof_${TAG}_train.log
head -40 gpurun_out/prof_${TAG}_train.md
