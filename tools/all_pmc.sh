#!/bin/bash
# One PMC pass over ALL kernels of a 340x510 step: vector-memory instruction counts, LDS conflicts, parked time.
# usage (GPU box): bash tools/all_pmc.sh <tag>
set -e
TAG=${1:-x}
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/apmc_${TAG} -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/apmc_${TAG}.log 2>&1
cd $R
python3 tools/pmc_summary.py gpurun_out/apmc_${TAG} > gpurun_out/apmc_${TAG}_summary.txt
rm -rf gpurun_out/apmc_${TAG}
python3 - <<PY
import re
blocks, cur = [], None
for ln in open("gpurun_out/apmc_${TAG}_summary.txt"):
    m = re.match(r"^(.*?) dispatches (\d+)", ln)
    if m:
        cur = {"name": m.group(1).replace("(anonymous namespace)::", "")[:60], "n": int(m.group(2))}
        blocks.append(cur)
    elif cur is not None and ln.strip():
        k, v = ln.split()[:2]
        cur[k] = float(v)
blocks.sort(key=lambda b: -b.get("SQ_BUSY_CYCLES", 0) * b["n"])
print(f"{'kernel':60s} {'n':>5s} {'busyMcyc':>9s} {'VMEM_RD':>10s} {'VMEM_WR':>10s} {'parked':>7s} {'ldsconf':>7s}")
for b in blocks[:40]:
    wc = max(b.get("SQ_WAVE_CYCLES", 1), 1)
    print(f"{b['name']:60s} {b['n']:5d} {b.get('SQ_BUSY_CYCLES', 0) / 32e6:9.3f} {b.get('SQ_INSTS_VMEM_RD', 0):10.0f} {b.get('SQ_INSTS_VMEM_WR', 0):10.0f} "
          f"{b.get('SQ_WAIT_ANY', 0) / wc:7.2f} {b.get('SQ_LDS_BANK_CONFLICT', 0) / max(b.get('SQ_LDS_IDX_ACTIVE', 1), 1):7.2f}")
PY
