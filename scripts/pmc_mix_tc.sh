#!/bin/bash
# Instruction mix / SQ utilisation of the fused TagContinuous tick (separate rocprofv3 --pmc passes).
# Run on the GPU box; prints one line per counter (average per launch).
cd "$(dirname "$0")/.."
R=$PWD
cd /tmp && export TMPDIR=/tmp
if [ "$1" = "list" ]; then rocprofv3 -L 2>/dev/null | grep -o "SQ_INSTS[A-Z0-9_]*\|SQ_ACTIVE[A-Z0-9_]*\|SQ_WAIT[A-Z0-9_]*\|SQ_INST_CYCLES[A-Z0-9_]*\|SQ_VALU[A-Z0-9_]*" | sort -u | tr '\n' ' '; echo; exit 0; fi
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F64" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM"; do
  d=/tmp/pmc_mix; rm -rf $d
  rocprofv3 --kernel-trace --pmc $grp -d $d -o pmc -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
  db=$(find $d -name "*.db" | head -1)
  for c in $grp; do
    python $R/scripts/rocpd_summary.py pmc $db $c 2>/dev/null | python -c "
import json,sys
try:
    for r in json.load(sys.stdin):
        if 'Tick' in r['kernel']: print('%-28s avg=%.5g' % (r['counter'], r['avg']))
except Exception as e: print('$c: n/a')"
  done
done
