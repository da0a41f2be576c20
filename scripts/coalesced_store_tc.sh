#!/bin/bash
# Timing experiment: would fully coalesced observation stores (WD_TC_ABLATE bit 128: same bytes, wrong
# layout) change the picture for the store policies (WD_TC_OBS_STORE 0 plain, 1 nt, 3 agent-scope write-through)?
set -e
cd "$(dirname "$0")/.."
mkdir -p build/store
FLAGS="--offload-arch=gfx950 --genco -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math"
for v in ${1:-0:0 128:0 128:3 128:1}; do
  IFS=: read ab st <<< "$v"
  out=build/store/wd_kernels_ab${ab}_st$st.hsaco
  hipcc $FLAGS -DWD_TC_ABLATE=$ab -DWD_TC_OBS_STORE=$st warp_drive_amd/csrc/kernels/wd_kernels.hip -o $out
  echo -n "ablate=$ab store=$st : "
  WD_HSACO=$PWD/$out python bench.py --steps 1000 --warmup 100 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms_per_step=%.4f kernel_us=%.2f' % (d['ms_per_step'], r['avg_kernel_us']))"
done
