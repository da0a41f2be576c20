#!/bin/bash
# Instruction mix / SQ utilisation of the fused TagContinuous tick (separate rocprofv3 --pmc passes,
# --kernel-trace only: gpurun refuses --pmc together with other trace domains).  Every pass runs WHOLE
# episodes (bench.py --profile-episodes), so a per-dispatch average is the episode average.
# Run on the GPU box; writes gpurun_out/profiles/<tag>_pmc_mix.txt and gpurun_out/profiles/pmc_mix.json
# (keyed by the code object's sha256; bench.py's roofline_valu reads profiles/pmc_mix.json).
# usage: scripts/pmc_mix_tc.sh <tag> [num_envs]
TAG=${1:-r03}; E=${2:-2000}
cd "$(dirname "$0")/.."
R=$PWD
O=$R/gpurun_out/profiles
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
OUT=$O/${TAG}_pmc_mix.txt
: > $OUT
echo "# rocprofv3 --kernel-trace --pmc <group> -- python bench.py --profile-episodes 1 --num-envs $E   (one line per counter: average per launch over one whole episode)" >> $OUT
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F64" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE"; do
  d=/tmp/pmc_mix; rm -rf $d
  rocprofv3 --kernel-trace --pmc $grp -d $d -o pmc -- python $R/bench.py --profile-episodes 1 --num-envs $E > /dev/null 2>&1
  db=$(find $d -name "*.db" | head -1)
  for c in $grp; do
    python $R/scripts/rocpd_summary.py pmc $db $c 2>/dev/null | python -c "
import json,sys
try:
    for r in json.load(sys.stdin):
        if 'Tick' in r['kernel']: print('%-28s avg=%.6g kernel=%s launches=%s' % (r['counter'], r['avg'], r['kernel'], r.get('dispatches', '?')))
except Exception as e: print('$c: n/a')" >> $OUT
  done
done
python - $OUT $E $R <<'PY' > $O/pmc_mix.json
import hashlib, json, re, sys
out, E, R = sys.argv[1], int(sys.argv[2]), sys.argv[3]
c, kernel = {}, None
for line in open(out):
    m = re.match(r"(\S+)\s+avg=(\S+) kernel=(\S+)", line)
    if m:
        c[m.group(1)] = float(m.group(2)); kernel = m.group(3).replace('.kd', '')
sys.path.insert(0, R)
from warp_drive_amd.managers import hip_driver
sha = hip_driver.code_object_sha256(kernel)  # the object that holds the kernel
rec = {"kernel": kernel, "num_envs": E, "full_obs": False, "hsaco_sha256": sha, "counters_per_launch": c,
       "note": "rocprofv3 --pmc passes over one whole 500-tick episode each (scripts/pmc_mix_tc.sh); SQ_WAVE_CYCLES / "
               "SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md)"}
if c.get("SQ_BUSY_CYCLES") and c.get("GRBM_GUI_ACTIVE"):
    rec["gui_active_cycles_per_launch"] = c["GRBM_GUI_ACTIVE"]
print(json.dumps(rec, indent=1))
PY
cat $OUT
