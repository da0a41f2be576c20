#!/bin/bash
# geometry experiment: replicas per block of HipTagContinuousStep (timing only)
cd "$(dirname "$0")/.."
for mt in 128 256 320 512 640 1024; do
  echo "== max_threads=$mt"
  WD_TC_MAX_THREADS=$mt python bench.py --steps 500 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms_per_step=%.4f step_kernel_us=%.2f' % (d['ms_per_step'], r['avg_kernel_us']))"
done
