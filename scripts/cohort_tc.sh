#!/bin/bash
# Timing experiment: start cohorts of HipTagContinuousTick (WD_TC_COHORTS / _SHIFT / _NS).
# usage (GPU box): scripts/cohort_tc.sh "N:SHIFT:NS ..."
set -e
cd "$(dirname "$0")/.."
mkdir -p build/cohort
FLAGS="--offload-arch=gfx950 --genco -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math"
for v in ${1:-1:8:0 2:8:4000}; do
  IFS=: read n sh ns <<< "$v"
  out=build/cohort/wd_kernels_c${n}_${sh}_${ns}.hsaco
  hipcc $FLAGS -DWD_TC_COHORTS=$n -DWD_TC_COHORT_SHIFT=$sh -DWD_TC_COHORT_NS=$ns warp_drive_amd/csrc/kernels/wd_kernels.hip -o $out
  echo -n "cohorts=$n shift=$sh ns=$ns : "
  WD_HSACO=$PWD/$out python bench.py --steps 1000 --warmup 100 --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms_per_step=%.4f kernel_us=%.2f' % (d['ms_per_step'], r['avg_kernel_us']))"
done
