#!/bin/bash
# Fused tick at several replica counts (run on the GPU box): one line per count --
#   num_envs  us_per_step  kernel_us  M_env_steps_per_s  roofline_frac
# usage: scripts/env_sweep.sh [num_envs ...] > gpurun_out/profiles/<tag>_env_sweep.txt
cd "$(dirname "$0")/.."
for E in ${@:-500 1000 1500 2000 3000 4000 8000 16000}; do
  python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --num-envs $E 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print($E, round(d['ms_per_step'] * 1000, 2), round(r['avg_kernel_us'], 2), round(d['value'] / 1e6, 2), round(r['frac'], 3))"
done
