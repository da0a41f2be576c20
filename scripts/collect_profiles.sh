#!/bin/bash
# Collect the rocprofv3 evidence kept under profiles/ (run on the GPU box; results land in
# gpurun_out/profiles/, copy them into profiles/ afterwards).  Kernel trace and PMC counters are
# separate runs; FETCH_SIZE and WRITE_SIZE are separate --pmc passes (MI355X_MICROARCH.md, HBM section).
# usage: scripts/collect_profiles.sh <tag>      e.g. r01_final2
set -e
TAG=${1:-r01}
cd "$(dirname "$0")/.."
R=$PWD
O=$R/gpurun_out/profiles
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
S=$O/${TAG}_kernel_trace_stats.txt
P=$O/${TAG}_pmc_hbm.txt
: > $S; : > $P
for mode in "" "--unfused"; do
  d=/tmp/prof_kt${mode:+_unfused}; rm -rf $d
  rocprofv3 --kernel-trace --stats -d $d -o kt -- python $R/bench.py --steps 1000 --warmup 100 --no-cpu-baseline $mode > $O/${TAG}_bench${mode:+_unfused}.json 2>/dev/null
  echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 1000 --warmup 100 --no-cpu-baseline $mode" >> $S
  python $R/scripts/rocpd_summary.py kernel $(find $d -name "*.db" | head -1) >> $S
  echo >> $S
  for c in FETCH_SIZE WRITE_SIZE; do
    d=/tmp/prof_$c${mode:+_unfused}; rm -rf $d
    rocprofv3 --kernel-trace --pmc $c -d $d -o pmc -- python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline $mode > /dev/null 2>&1
    echo "# rocprofv3 --kernel-trace --pmc $c -- python bench.py --steps 40 --warmup 10 --no-cpu-baseline $mode" >> $P
    python $R/scripts/rocpd_summary.py pmc $(find $d -name "*.db" | head -1) $c >> $P
  done
done
d=/tmp/prof_sq; rm -rf $d
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d $d -o pmc -- python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
echo "# rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -- python bench.py --steps 40 --warmup 10 --no-cpu-baseline" >> $P
for c in SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES; do python $R/scripts/rocpd_summary.py pmc $(find $d -name "*.db" | head -1) $c >> $P; done
cat $S
