#!/bin/bash
# A/B two code objects on ANY bench.py workload in one gpurun call, interleaved.
# usage: scripts/ab_bench_any.sh <a.hsaco|default> <b.hsaco|default> <rounds> <bench args...>
cd "$(dirname "$0")/.."
A=$1; B=$2; R=$3; shift 3
one() {
  if [ "$1" = "default" ]; then unset WD_HSACO; else export WD_HSACO=$PWD/$1; fi
  python bench.py --no-cpu-baseline --no-spread "${@:2}" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['config'].get('ticks_per_launch',1)
print('%-28s us_per_launch=%.2f us_per_tick=%.3f value=%.4g' % ('$1', d['ms_per_step']*1e3, d['ms_per_step']*1e3/t, d['value']))"
}
for i in $(seq $R); do one $A "$@"; one $B "$@"; done
