#!/bin/bash
# configs[4] ceiling run: Cartpole, 100 000 replicas, 1 tick per launch vs T ticks per launch with every tick
# recorded in the trainer's [T, E, ...] batch tensors; WRITE_SIZE / FETCH_SIZE PMC of the T = 50 launch.
# Run on the GPU box; output gpurun_out/profiles/<tag>_cartpole_ceiling.txt
TAG=${1:-r03}
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/profiles; mkdir -p $O
OUT=$O/${TAG}_cartpole_ceiling.txt; : > $OUT
line() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']
print('ticks_per_launch=%-3d value %.4g env-steps/s  ms_per_launch %.4f  kernel %.2f us  bytes/launch %.4g  achieved %.0f GB/s  frac %.3f' % (d['config']['ticks_per_launch'], d['value'], d['ms_per_step'], r['avg_kernel_us'], r['algorithmic_bytes_per_launch'], r['achieved'], r['frac']))" >> $OUT; }
echo "# python bench.py --workload cartpole [--ticks-per-launch T] --no-cpu-baseline   (plain runs)" >> $OUT
python bench.py --workload cartpole --steps 2000 --warmup 100 --no-cpu-baseline > /tmp/cp.json 2>/dev/null; cp /tmp/cp.json $O/${TAG}_bench_cartpole_T1.json; line /tmp/cp.json
for T in 10 50 100; do python bench.py --workload cartpole --ticks-per-launch $T --steps 300 --warmup 50 --no-cpu-baseline > /tmp/cp.json 2>/dev/null; cp /tmp/cp.json $O/${TAG}_bench_cartpole_T$T.json; line /tmp/cp.json; done
echo "# the same with more replicas than configs[4] asks for (100 000 replicas are 1.5 wavefronts per SIMD): --num-envs 400000 / 1600000, T = 50" >> $OUT
for E in 400000 1600000; do python bench.py --workload cartpole --ticks-per-launch 50 --num-envs $E --steps 100 --warmup 20 --no-cpu-baseline --no-spread > /tmp/cp.json 2>/dev/null; cp /tmp/cp.json $O/${TAG}_bench_cartpole_T50_E$E.json; line /tmp/cp.json; done
cd /tmp && export TMPDIR=/tmp
for c in WRITE_SIZE FETCH_SIZE; do
  rm -rf /tmp/cp_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/cp_$c -o pmc -- python $R/bench.py --workload cartpole --ticks-per-launch 50 --steps 40 --warmup 10 --no-cpu-baseline --no-spread > /dev/null 2>&1
  echo "# rocprofv3 --kernel-trace --pmc $c -- python bench.py --workload cartpole --ticks-per-launch 50 --steps 40 --warmup 10 --no-cpu-baseline --no-spread   (KB per launch)" >> $OUT
  python $R/scripts/rocpd_summary.py pmc $(find /tmp/cp_$c -name "*.db" | head -1) $c | python -c "
import json,sys
for r in json.load(sys.stdin):
    if 'CartPole' in r['kernel']: print('%s %s avg=%.1f KB over %d launches' % (r['kernel'], r['counter'], r['avg'], r['dispatches']))" >> $OUT
done
cat $OUT
