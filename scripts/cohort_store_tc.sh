#!/bin/bash
# Timing experiment: start cohorts x observation store policy.  usage: cohort_store_tc.sh "STORE:N:SHIFT:NS ..."
set -e
cd "$(dirname "$0")/.."
mkdir -p build/cohort
FLAGS="--offload-arch=gfx950 --genco -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math"
for v in $1; do
  IFS=: read st n sh ns <<< "$v"
  out=build/cohort/wd_kernels_s${st}_c${n}_${sh}_${ns}.hsaco
  hipcc $FLAGS -DWD_TC_OBS_STORE=$st -DWD_TC_COHORTS=$n -DWD_TC_COHORT_SHIFT=$sh -DWD_TC_COHORT_NS=$ns warp_drive_amd/csrc/kernels/wd_kernels.hip -o $out
  echo -n "store=$st cohorts=$n shift=$sh ns=$ns : "
  WD_HSACO=$PWD/$out python bench.py --steps 1000 --warmup 100 --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms_per_step=%.4f kernel_us=%.2f' % (d['ms_per_step'], r['avg_kernel_us']))"
done
