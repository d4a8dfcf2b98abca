#!/bin/bash
# Timing-only ablations of the fused token-chain kernel (results are wrong by construction): builds probe copies of the library
# with -DFFSR_TOK_ABL=<bits> and times tools/tok_bench.py with each.  bits: 1 no activation math, 2 no fragment reads from LDS,
# 4 no barriers / waits, 8 no LDS-DMA fills, 16 no MFMAs.   usage (GPU box): bash tools/tok_ablate.sh "0 1 2 4 8 16 ..." [K H]
set -e
R=$PWD
C=$R/image-super-resolution_amd/csrc
mkdir -p /tmp/tokabl
for b in $1; do
  /opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++17 -Wno-unused-result -DFFSR_TOK_ABL=$b -c $C/ffsr_tok.hip -o /tmp/tokabl/tok_$b.o
  objs=$(ls $C/*.o | grep -v ffsr_tok.o)
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs /tmp/tokabl/tok_$b.o -o /tmp/tokabl/lib_$b.so
  echo "== ablation bits $b"
  FFSR_LIB=/tmp/tokabl/lib_$b.so FFSR_TOK_ONLY=1 python3 $R/tools/tok_bench.py ${2:-180} ${3:-360} 2>&1 | grep -E "waves|fused"
done
