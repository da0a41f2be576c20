#!/bin/bash
# Timing experiment: store policy of the observation rows in HipTagContinuousTick
# (WD_TC_OBS_STORE: 0 plain, 1 non-temporal, 2 system-scope write-through).  Run on the GPU box.
set -e
cd "$(dirname "$0")/.."
mkdir -p build/store
FLAGS="--offload-arch=gfx950 --genco -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math"
for v in ${VARIANTS:-0 1 2}; do
  out=build/store/wd_kernels_st$v.hsaco
  hipcc $FLAGS -DWD_TC_OBS_STORE=$v warp_drive_amd/csrc/kernels/wd_kernels.hip -o $out
  echo "== obs store policy $v"
  WD_HSACO=$PWD/$out python bench.py --steps 1000 --warmup 100 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms_per_step=%.4f kernel_us=%.2f' % (d['ms_per_step'], r['avg_kernel_us']))"
done
