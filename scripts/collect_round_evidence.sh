#!/bin/bash
# Everything kept under profiles/ for one round, in ONE call on the GPU box (~14 min): output under
# gpurun_out/profiles/, copy what is to be judged into profiles/.   usage: scripts/collect_round_evidence.sh r04
# (most important first: a call that runs out of GPU budget loses the explanatory items at the end)
TAG=${1:-r04}
cd "$(dirname "$0")/.."
R=$PWD
O=$R/gpurun_out/profiles
mkdir -p $O
# 1. kernel-trace stats + FETCH / WRITE PMC of the fused tick at 2000 / 8000 / 16000 replicas (+ --unfused at 2000)
bash scripts/collect_profiles.sh $TAG 2000 8000 16000 > /dev/null 2>&1
# 2. instruction mix / SQ counters at the headline size
bash scripts/pmc_mix_tc.sh $TAG 2000 > /dev/null 2>&1
# 3. plain bench lines: default, the driver's shape, the worst case (nobody is ever tagged)
python bench.py > $O/${TAG}_bench_default.json 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_driver_shape.json 2>/dev/null
python bench.py --no-tags --steps 1000 --warmup 100 > $O/${TAG}_bench_notags.json 2>/dev/null
# 4. side workloads: plain lines + kernel trace + FETCH / WRITE PMC
bash scripts/collect_side_profiles.sh $TAG > /dev/null 2>&1
# 5. phase profiles (build/variants/prof.hsaco = the product source with -DWD_TC_PROBES)
for t in 20 300; do python experiments/phase_profile.py prof 2000 $t > $O/${TAG}_phase_profile_E2000_t$t.txt 2>&1; done
python experiments/phase_profile.py prof 256 20 1000 > $O/${TAG}_phase_profile_N1005_E256_t20.txt 2>&1
# 6. the tick along an episode
python scripts/episode_profile.py > $O/${TAG}_episode_profile.txt 2>&1
# 7. T-tick rollouts of the side workloads
for t in 1 10 50 100; do python bench.py --workload cartpole --ticks-per-launch $t --no-cpu-baseline > $O/${TAG}_bench_cartpole_T$t.json 2>/dev/null; done
python bench.py --workload cartpole --ticks-per-launch 50 --num-envs 1600000 --steps 200 --warmup 20 --no-cpu-baseline > $O/${TAG}_bench_cartpole_T50_E1600000.json 2>/dev/null
for t in 50 200; do python bench.py --workload tag_gridworld --ticks-per-launch $t --no-cpu-baseline --steps 500 --warmup 50 > $O/${TAG}_bench_gridworld_E1000_T$t.json 2>/dev/null; done
python bench.py --workload tag_gridworld --num-envs 100000 --ticks-per-launch 50 --no-cpu-baseline --steps 100 --warmup 10 > $O/${TAG}_bench_gridworld_E100000_T50.json 2>/dev/null
bash scripts/pmc_mix.sh gpurun_out/profiles/${TAG}_pmc_mix_cartpole_T50.txt CartPole --workload cartpole --ticks-per-launch 50 --steps 20 --warmup 5 > /dev/null 2>&1
bash scripts/pmc_mix.sh gpurun_out/profiles/${TAG}_pmc_mix_gridworld_T50.txt GridWorld --workload tag_gridworld --ticks-per-launch 50 --steps 20 --warmup 5 > /dev/null 2>&1
# 8. trainer: rollout + iteration at configs[2], Cartpole rollout with the policy inside the kernel
python scripts/rollout_timing.py > $O/${TAG}_rollout_timing.txt 2>&1
python scripts/cartpole_rollout_timing.py > $O/${TAG}_cartpole_rollout_timing.txt 2>&1
python scripts/gridworld_rollout_timing.py > $O/${TAG}_gridworld_rollout_timing.txt 2>&1
# 9. beyond the Infinity Cache: instruction mix and phase profile at 16000 replicas
cp $O/pmc_mix.json $O/pmc_mix_E2000.json
bash scripts/pmc_mix_tc.sh ${TAG}_E16000 16000 > /dev/null 2>&1; cp $O/pmc_mix.json $O/pmc_mix_E16000.json; cp $O/pmc_mix_E2000.json $O/pmc_mix.json
python experiments/phase_profile.py prof 16000 300 > $O/${TAG}_phase_profile_E16000_t300.txt 2>&1
ls $O | wc -l
