#!/bin/bash
# Collect the rocprofv3 evidence kept under profiles/ (run on the GPU box; results land in
# gpurun_out/profiles/, copy them into profiles/ afterwards).  Kernel trace and PMC counters are
# separate runs; FETCH_SIZE and WRITE_SIZE are separate --pmc passes (MI355X_MICROARCH.md, HBM section).
# usage: scripts/collect_profiles.sh <tag> [num_envs ...]      e.g. r02 2000 8000 16000
set -e
TAG=${1:-r02}; shift || true
ENVS=${@:-2000}
cd "$(dirname "$0")/.."
R=$PWD
O=$R/gpurun_out/profiles
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
S=$O/${TAG}_kernel_trace_stats.txt
P=$O/${TAG}_pmc_hbm.txt
J=$O/${TAG}_pmc_traffic.json
: > $S; : > $P
echo "{" > $J
first=1
for E in $ENVS; do
  for mode in "" "--unfused"; do
    [ -n "$mode" ] && [ "$E" != "2000" ] && continue
    d=/tmp/prof_kt_$E${mode:+_unfused}; rm -rf $d
    rocprofv3 --kernel-trace --stats -d $d -o kt -- python $R/bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-spread --num-envs $E $mode > $O/${TAG}_bench_E$E${mode:+_unfused}.json 2>/dev/null
    echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-spread --num-envs $E $mode   (2000 launches + pre-roll: whole episodes, so the average is the episode average)" >> $S
    python $R/scripts/rocpd_summary.py kernel $(find $d -name "*.db" | head -1) >> $S
    echo >> $S
    for c in FETCH_SIZE WRITE_SIZE; do
      d=/tmp/prof_${c}_$E${mode:+_unfused}; rm -rf $d
      if [ -z "$mode" ]; then PA="--profile-episodes 1"; else PA="--steps 40 --warmup 10 --no-cpu-baseline --no-spread"; fi
      rocprofv3 --kernel-trace --pmc $c -d $d -o pmc -- python $R/bench.py $PA --num-envs $E $mode > /dev/null 2>&1
      echo "# rocprofv3 --kernel-trace --pmc $c -- python bench.py $PA --num-envs $E $mode   (fused tick: one whole episode, the average is the episode average)" >> $P
      python $R/scripts/rocpd_summary.py pmc $(find $d -name "*.db" | head -1) $c >> $P
      cp $(find $d -name "*.db" | head -1) /tmp/last_$c.db
    done
    if [ -z "$mode" ]; then
      [ $first = 1 ] || echo "," >> $J
      first=0
      python - $E /tmp/last_FETCH_SIZE.db /tmp/last_WRITE_SIZE.db $R >> $J <<'PY'
import hashlib, json, sys
sys.path.insert(0, sys.argv[4] + "/scripts")
from rocpd_summary import pmc_stats
E = int(sys.argv[1])
f = [r for r in pmc_stats(sys.argv[2], "FETCH_SIZE") if "Tick" in r["kernel"]][0]
w = [r for r in pmc_stats(sys.argv[3], "WRITE_SIZE") if "Tick" in r["kernel"]][0]
sys.path.insert(0, sys.argv[4])
from warp_drive_amd.managers import hip_driver
sha = hip_driver.code_object_sha256(f["kernel"].replace(".kd", ""))  # the object that holds the kernel
rec = {"kernel": f["kernel"], "num_envs": E, "full_obs": False, "hsaco_sha256": sha,
       "fetch_size_kb": f["avg"], "write_size_kb": w["avg"],
       "hbm_bytes_per_launch": (2 * f["avg"] + w["avg"]) * 1024,
       "correction": "2*FETCH_SIZE + WRITE_SIZE (gfx950 FETCH_SIZE halves coalesced reads)"}
print(f'"E{E}": ' + json.dumps(rec, indent=1))
PY
    fi
  done
done
echo "}" >> $J
cat $S
cat $J
