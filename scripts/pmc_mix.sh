#!/bin/bash
# Instruction mix / SQ utilisation counters of ANY bench.py workload (generalises pmc_mix_tc.sh): separate
# rocprofv3 --pmc passes (--kernel-trace only: gpurun refuses --pmc together with other trace domains), one line per
# counter = average per launch of the kernel whose name contains <kernel-substring>.
# usage: scripts/pmc_mix.sh <out-file> <kernel-substring> <bench args...>
#   e.g. scripts/pmc_mix.sh gpurun_out/profiles/r04_pmc_mix_cartpole_T50.txt CartPole --workload cartpole --ticks-per-launch 50 --steps 20 --warmup 5
OUT=$1; KSUB=$2; shift 2
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p "$(dirname "$R/$OUT")"
OUT=$R/$OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT
echo "# rocprofv3 --kernel-trace --pmc <group> -- python bench.py $* --no-cpu-baseline --no-spread   (average per launch of *$KSUB*)" >> $OUT
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F64" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_EXP_GDS SQ_INSTS_FLAT SQ_IFETCH"; do
  d=/tmp/pmc_mix_any; rm -rf $d
  rocprofv3 --kernel-trace --pmc $grp -d $d -o pmc -- python $R/bench.py "$@" --no-cpu-baseline --no-spread > /dev/null 2>&1
  db=$(find $d -name "*.db" | head -1)
  for c in $grp; do
    python $R/scripts/rocpd_summary.py pmc $db $c 2>/dev/null | python -c "
import json,sys
try:
    for r in json.load(sys.stdin):
        if '$KSUB' in r['kernel']: print('%-28s avg=%.6g kernel=%s launches=%s' % (r['counter'], r['avg'], r['kernel'], r.get('dispatches', '?')))
except Exception as e: print('$c: n/a')" >> $OUT
  done
done
cat $OUT
