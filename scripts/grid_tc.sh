#!/bin/bash
# grid-size experiment: fewer blocks than replica groups => each block loops over groups and
# the observation stores of group g overlap the neighbour search of group g+1 (timing only)
cd "$(dirname "$0")/.."
for g in ${GRIDS:-1000 768 512 500 334 256}; do
  echo "== grid=$g max_threads=${WD_TC_MAX_THREADS:-256}"
  WD_TC_GRID=$g python bench.py --steps 500 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms_per_step=%.4f step_kernel_us=%.2f' % (d['ms_per_step'], r['avg_kernel_us']))"
done
