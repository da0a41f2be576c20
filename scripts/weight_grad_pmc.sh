#!/bin/bash
# SQ counters of the update's bf16x3 kernels (weight gradients, input gradient) (separate rocprofv3 passes); workload = scripts/weight_grad_timing.py.
# Run on the GPU box:  bash scripts/weight_grad_pmc.sh > gpurun_out/weight_grad_pmc.txt
cd "$(dirname "$0")/.."
R=$PWD
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  d=/tmp/pmc_wg; rm -rf $d
  timeout 200 rocprofv3 --kernel-trace --pmc $grp -d $d -o pmc -- python $R/scripts/weight_grad_timing.py kernels-only > /dev/null 2>&1
  db=$(find $d -name "*.db" | head -1)
  for c in $grp; do
    python $R/scripts/rocpd_summary.py pmc $db $c | python -c "
import json,sys
for r in json.load(sys.stdin):
    if 'Bx3' in r['kernel']: print('%-28s %-28s avg=%.4g' % (r['counter'], r['kernel'], r['avg']))"
  done
done
