#!/bin/bash
# Dynamic instruction counts of HipTagContinuousTick per phase: SQ_INSTS_* for the ablation builds
# (WD_TC_ABLATE: 1 no neighbour search, 2 no gather, 4 stop after search part A, 8 after part B).
set -e
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p build/ablate gpurun_out/pmc_ablate
FLAGS="--offload-arch=gfx950 --genco -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math"
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-0 1 2 3 4 8}; do
  out=$R/build/ablate/wd_kernels_ab$v.hsaco
  hipcc $FLAGS -DWD_TC_ABLATE=$v $R/warp_drive_amd/csrc/kernels/wd_kernels.hip -o $out
  d=$R/gpurun_out/pmc_ablate/ab$v
  rm -rf $d
  WD_HSACO=$out rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d $d -o pmc -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
  db=$(find $d -name "*.db" | head -1)
  echo "== ablate=$v"
  for c in SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES; do
    python $R/scripts/rocpd_summary.py pmc $db $c | python -c "
import json,sys
for r in json.load(sys.stdin):
    if 'Tick' in r['kernel']: print('   %-14s avg=%.0f' % (r['counter'], r['avg']))"
  done
done
