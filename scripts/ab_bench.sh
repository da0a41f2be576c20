#!/bin/bash
# A/B two code objects in ONE gpurun call, interleaved (boxes of the pool differ by up to 15 % in clock).
# usage: scripts/ab_bench.sh <a.hsaco|default> <b.hsaco> [rounds] [bench args...]
# prints per run: us per step of the timed region, the average launch over whole episodes, the full-load window
cd "$(dirname "$0")/.."
A=$1; B=$2; R=${3:-3}; shift 3 2>/dev/null
one() {
  if [ "$1" = "default" ]; then unset WD_HSACO; else export WD_HSACO=$PWD/$1; fi
  python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-spread "${@:2}" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; e=d['episode']
print('%-28s us_per_step=%.2f kernel_us=%.2f full_load_us=%s min=%.2f max=%.2f' % ('$1', d['ms_per_step']*1e3, r['avg_kernel_us'], ('%.2f' % r['full_load_us']) if r.get('full_load_us') else 'n/a', e['us_per_tick_min'], e['us_per_tick_max']))"
}
for i in $(seq $R); do one $A "$@"; one $B "$@"; done
