#!/bin/bash
# rocprofv3 kernel-trace stats of the side workloads quoted in DESIGN.md (full observations,
# TagGridWorld, Cartpole).  Run on the GPU box; output under gpurun_out/profiles/.
set -e
TAG=${1:-r01}
cd "$(dirname "$0")/.."
R=$PWD
O=$R/gpurun_out/profiles
mkdir -p $O
S=$O/${TAG}_side_kernel_trace_stats.txt
: > $S
cd /tmp && export TMPDIR=/tmp
run() {
  d=/tmp/prof_side; rm -rf $d
  rocprofv3 --kernel-trace --stats -d $d -o kt -- python $R/bench.py --no-cpu-baseline "$@" > /tmp/side.json 2>/dev/null
  echo "# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline $*" >> $S
  python $R/scripts/rocpd_summary.py kernel $(find $d -name "*.db" | head -1) | head -6 >> $S
  python -c "
import json
d=[json.loads(l) for l in open('/tmp/side.json') if l.startswith('{')][-1]
print('# bench.py: value=%.4g %s, ms_per_step=%.4f, roofline achieved %.0f GB/s frac %.3f (%s, avg %.2f us)' % (d['value'], d['unit'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['kernel'], d['roofline']['avg_kernel_us']))" >> $S
  echo >> $S
}
run --full-obs --steps 300 --warmup 30
run --workload tag_gridworld --steps 2000 --warmup 100
run --workload tag_gridworld --num-envs 100000 --steps 1000 --warmup 100
run --workload cartpole --steps 2000 --warmup 100
cat $S
