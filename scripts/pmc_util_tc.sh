#!/bin/bash
# SQ utilisation counters of the fused TagContinuous tick (separate rocprofv3 passes; quad-cycle units
# for SQ_ACTIVE_* / SQ_WAIT_* / SQ_WAVE_CYCLES, see MI355X_MICROARCH.md).  Run on the GPU box.
cd "$(dirname "$0")/.."
R=$PWD
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_INSTS_VALU_TRANS_F32" \
           "SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64"; do
  d=/tmp/pmc_util; rm -rf $d
  rocprofv3 --kernel-trace --pmc $grp -d $d -o pmc -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
  db=$(find $d -name "*.db" | head -1)
  for c in $grp; do
    python $R/scripts/rocpd_summary.py pmc $db $c | python -c "
import json,sys
for r in json.load(sys.stdin):
    if 'Tick' in r['kernel']: print('%-28s avg=%.4g' % (r['counter'], r['avg']))"
  done
done
