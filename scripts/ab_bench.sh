#!/bin/bash
# A/B two code objects in ONE gpurun call, interleaved (boxes of the pool differ by up to 15 % in clock).
# usage: scripts/ab_bench.sh <a.hsaco|default> <b.hsaco> [rounds] [bench args...]
cd "$(dirname "$0")/.."
A=$1; B=$2; R=${3:-3}; shift 3 2>/dev/null
one() {
  if [ "$1" = "default" ]; then unset WD_HSACO; else export WD_HSACO=$PWD/$1; fi
  python bench.py --steps 2000 --warmup 200 --no-cpu-baseline "${@:2}" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s ms_per_step=%.5f kernel_us=%.2f' % ('$1', d['ms_per_step'], d['roofline']['avg_kernel_us']))"
}
for i in $(seq $R); do one $A "$@"; one $B "$@"; done
