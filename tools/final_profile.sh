#!/bin/bash
# Round-end evidence on the GPU box: rocprofv3 kernel-trace stats (default + sequential experts) and the two PMC passes
# (FETCH_SIZE, WRITE_SIZE; each in its own run, no other trace domains).  usage: bash tools/final_profile.sh <tag>
set -e
TAG=${1:-x}
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_conc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_${TAG}_conc.log 2>&1
echo "conc done"
FFSR_CONCURRENT_EXPERTS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_seq -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_${TAG}_seq.log 2>&1
echo "seq done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $R/gpurun_out/pmc_${TAG}_fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $R/gpurun_out/pmc_${TAG}_write.log 2>&1
echo "write done"
cd $R
# keep only the small summaries (gpurun merges at most 64 MiB back)
for k in conc seq; do
  f=$(ls gpurun_out/prof_${TAG}_$k/*/*_kernel_stats.csv | head -1)
  cp $f gpurun_out/prof_${TAG}_${k}_kernel_stats.csv
done
python3 tools/pmc_traffic.py gpurun_out/pmc_${TAG}_fetch gpurun_out/pmc_${TAG}_write gpurun_out/pmc_${TAG}_traffic.json "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --kernel-trace -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras" > /dev/null
du -sh gpurun_out/* | sort -h | tail -8
rm -rf gpurun_out/prof_${TAG}_conc gpurun_out/prof_${TAG}_seq gpurun_out/pmc_${TAG}_fetch gpurun_out/pmc_${TAG}_write
