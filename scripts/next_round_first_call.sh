#!/bin/bash
# What the end of round 4 left READY TO MEASURE, as one GPU call (~6 min of box time).  Prepare here first (CPU, ~12 min):
#     python experiments/gw5_policy/run_parity.py build
#     python experiments/variants.py build constfold && python experiments/variants.py build prefilter_next
# then:  gpurun --timeout 600 -- 'bash scripts/next_round_first_call.sh'
# Output: gpurun_out/next/*.txt
cd "$(dirname "$0")/.."
O=gpurun_out/next
mkdir -p $O
# 1. the suite (incl. the two big-replica goldens added after the last GPU run of round 4)
(timeout 400 python -m pytest tests -m gpu -q -x 2>&1 | tail -15) > $O/gpu_tests.txt
# 2. the live-policy TagGridWorld rollout (experiments/gw5_policy/README.md): parity against the oracle + us per tick
for h in 32 64; do (timeout 60 python experiments/gw5_policy/run_parity.py $h 1000 20 2>&1 | tail -4) > $O/gw5_policy_H$h.txt; done
# 3. headline A/B: the BASELINE shape's sizes as compile-time constants (upper bound of what _N105-style entries give)
(timeout 120 python experiments/variants.py bench constfold 3 --no-spread 2>&1 | tail -3) > $O/ab_constfold.txt
# 4. big replicas: the provable (K + 1)-th-smallest radius for the prefiltered search
(timeout 150 python experiments/variants.py bench prefilter_next 2 --num-runners 500 --steps 500 --warmup 50 --no-spread 2>&1 | tail -3) > $O/ab_prefilter_next_505.txt
(timeout 200 python experiments/variants.py bench prefilter_next 1 --num-runners 1000 --steps 100 --warmup 10 --no-spread 2>&1 | tail -3) > $O/ab_prefilter_next_1005.txt
tail -n +1 $O/*.txt
(timeout 120 python bench.py 2>&1 | tail -1) > $O/bench_default.txt; tail -n +1 $O/bench_default.txt
