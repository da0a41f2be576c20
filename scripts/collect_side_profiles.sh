#!/bin/bash
# Evidence for the side workloads quoted in DESIGN.md / BASELINE.md section 4 (full observations, TagGridWorld,
# agent counts beyond the fast path): for every row the PLAIN bench.py JSON line (kept as
# <tag>_bench_<name>.json), the rocprofv3 kernel-trace stats and the FETCH_SIZE / WRITE_SIZE PMC passes
# (separate --pmc runs).  Run on the GPU box; output under gpurun_out/profiles/.
TAG=${1:-r03}
cd "$(dirname "$0")/.."
R=$PWD
O=$R/gpurun_out/profiles
mkdir -p $O
S=$O/${TAG}_side_workloads.txt
: > $S
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  python $R/bench.py --no-cpu-baseline "$@" > $O/${TAG}_bench_$name.json 2>/dev/null
  echo "## $name: python bench.py --no-cpu-baseline $*" >> $S
  python -c "
import json
d=[json.loads(l) for l in open('$O/${TAG}_bench_$name.json') if l.startswith('{')][-1]
r=d['roofline']
print('plain run: value=%.4g %s, ms_per_step=%.5f (spread %.5f..%.5f), %s avg %.2f us, algorithmic %.4g B/launch, achieved %.0f GB/s, frac %.3f' % (d['value'], d['unit'], d['ms_per_step'], d['ms_per_step_spread']['min'], d['ms_per_step_spread']['max'], r['kernel'], r['avg_kernel_us'], r['algorithmic_bytes_per_launch'], r['achieved'], r['frac']))" >> $S
  d=/tmp/prof_side; rm -rf $d
  rocprofv3 --kernel-trace --stats -d $d -o kt -- python $R/bench.py --no-cpu-baseline --no-spread "$@" > /dev/null 2>&1
  echo "rocprofv3 --kernel-trace --stats (same command + --no-spread):" >> $S
  python $R/scripts/rocpd_summary.py kernel $(find $d -name "*.db" | head -1) | head -4 >> $S
  for c in FETCH_SIZE WRITE_SIZE; do
    d=/tmp/prof_side_$c; rm -rf $d
    rocprofv3 --kernel-trace --pmc $c -d $d -o pmc -- python $R/bench.py --no-cpu-baseline --no-spread "$@" --steps 40 --warmup 10 > /dev/null 2>&1
    python $R/scripts/rocpd_summary.py pmc $(find $d -name "*.db" | head -1) $c | python -c "
import json,sys
rows=[r for r in json.load(sys.stdin) if r['kernel'].startswith('Hip')]
for r in rows[:1]: print('rocprofv3 --pmc %s (--steps 40 --warmup 10): %s avg=%.1f KB per launch over %d launches' % (r['counter'], r['kernel'], r['avg'], r['dispatches']))" >> $S
  done
  echo >> $S
}
run fullobs --full-obs --steps 300 --warmup 30
run gridworld_E1000 --workload tag_gridworld --steps 2000 --warmup 100
run gridworld_E100000 --workload tag_gridworld --num-envs 100000 --steps 1000 --warmup 100
run tc_150agents --num-runners 145 --steps 500 --warmup 50
run tc_505agents --num-runners 500 --steps 500 --warmup 50
run tc_1005agents --num-runners 1000 --steps 100 --warmup 10
cat $S
