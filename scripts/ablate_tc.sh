#!/bin/bash
# Build timing-only variants of the code object with phases of HipTagContinuousStep removed
# (WD_TC_ABLATE bits, see tag_continuous.hip) and time each with bench.py on the GPU box.
set -e
cd "$(dirname "$0")/.."
mkdir -p build/ablate
FLAGS="--offload-arch=gfx950 --genco -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math"
for v in ${VARIANTS:-0 1 2 3 16 64}; do
  out=build/ablate/wd_kernels_ab$v.hsaco
  [ -f $out ] || hipcc $FLAGS -DWD_TC_ABLATE=$v warp_drive_amd/csrc/kernels/wd_kernels.hip -o $out
  if [ "$1" = "run" ]; then
    echo "== ablate=$v"
    WD_HSACO=$PWD/$out python bench.py --steps ${STEPS:-500} --warmup 50 --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms_per_step=%.4f step_kernel_us=%.2f' % (d['ms_per_step'], r['avg_kernel_us']))"
  fi
done
