#!/bin/bash
# Round evidence (rounds 5 and 6) in ONE call on the GPU box (~8 min): output under gpurun_out/profiles/, copy what is to be judged into
# profiles/.  Every profiler run sits under `timeout` (a rocprofv3 that does not return otherwise eats the lease).
TAG=${1:-r05}
cd "$(dirname "$0")/.."
R=$PWD
O=$R/gpurun_out/profiles
mkdir -p $O
# 1. kernel-trace stats + FETCH / WRITE PMC of the fused tick at 2000 / 16000 replicas (+ --unfused at 2000)
timeout 400 bash scripts/collect_profiles.sh $TAG 2000 16000 > /dev/null 2>&1
# 2. instruction mix / SQ counters at the headline size
timeout 300 bash scripts/pmc_mix_tc.sh $TAG 2000 > /dev/null 2>&1
# 3. plain bench lines: default (with the CPU baseline), the driver's shape, the worst case (nobody is ever tagged)
timeout 120 python bench.py > $O/${TAG}_bench_default.json 2>/dev/null
timeout 120 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_driver_shape.json 2>/dev/null
timeout 120 python bench.py --no-tags --steps 1000 --warmup 100 --no-cpu-baseline > $O/${TAG}_bench_notags.json 2>/dev/null
timeout 120 python bench.py --full-obs --no-cpu-baseline --steps 500 --warmup 50 > $O/${TAG}_bench_fullobs.json 2>/dev/null
# 4. big replicas (the episode-average rate is in the line: value_episode_average)
timeout 200 python bench.py --num-runners 1000 --steps 100 --warmup 10 --no-cpu-baseline --no-spread > $O/${TAG}_bench_tc_1005agents.json 2>/dev/null
timeout 200 python bench.py --num-runners 500 --steps 500 --warmup 50 --no-cpu-baseline --no-spread > $O/${TAG}_bench_tc_505agents.json 2>/dev/null
# 5. side workloads
for t in 1 50; do timeout 100 python bench.py --workload cartpole --ticks-per-launch $t --no-cpu-baseline > $O/${TAG}_bench_cartpole_T$t.json 2>/dev/null; done
timeout 100 python bench.py --workload tag_gridworld --no-cpu-baseline > $O/${TAG}_bench_gridworld_E1000.json 2>/dev/null
for t in 50 200; do timeout 100 python bench.py --workload tag_gridworld --ticks-per-launch $t --no-cpu-baseline --steps 500 --warmup 50 > $O/${TAG}_bench_gridworld_E1000_T$t.json 2>/dev/null; done
timeout 100 python bench.py --workload tag_gridworld --num-envs 100000 --ticks-per-launch 50 --no-cpu-baseline --steps 100 --warmup 10 > $O/${TAG}_bench_gridworld_E100000_T50.json 2>/dev/null
# 6. the tick along an episode
timeout 120 python scripts/episode_profile.py > $O/${TAG}_episode_profile.txt 2>&1
# 7. trainer: rollout + iteration at configs[2]; TagGridWorld / Cartpole rollouts with the policies inside the kernel
timeout 300 python scripts/rollout_timing.py 2>&1 | grep -v "amdgpu.ids\|UserWarning\|over_agents" > $O/${TAG}_rollout_timing.txt
timeout 200 python scripts/gridworld_rollout_timing.py 2>&1 | grep -v "amdgpu.ids\|UserWarning\|over_agents" > $O/${TAG}_gridworld_rollout_timing.txt
timeout 200 python scripts/cartpole_rollout_timing.py 2>&1 | grep -v "amdgpu.ids\|UserWarning\|over_agents" > $O/${TAG}_cartpole_rollout_timing.txt
# 8. kernel view of the trainer (3 x (rollout of 50 ticks + update))
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/up; timeout 200 rocprofv3 --kernel-trace -d /tmp/up -o up -- python $R/scripts/update_profile.py float32 > /dev/null 2>&1
cd $R; db=$(find /tmp/up -name "*.db" | head -1)
[ -n "$db" ] && (echo "# rocprofv3 --kernel-trace -- python scripts/update_profile.py float32   (3 x (rollout of 50 ticks + update) at configs[2])"; python scripts/rocpd_summary.py kernel $db | head -40) > $O/${TAG}_update_kernels.txt
[ -n "$db" ] && (echo "# the kernels of one update in launch order (>= 100 us; gaps without a kernel >= 100 us): python scripts/update_timeline.py <db of the run above> 100"; python scripts/update_timeline.py $db 100) > $O/${TAG}_update_timeline.txt
# 9. the update's matrix-core kernels alone, against the framework GEMMs they replace (10 M rows of random data)
timeout 100 python scripts/weight_grad_timing.py 2>&1 | grep -v "amdgpu.ids" > $O/${TAG}_weight_grad_timing.txt
ls $O | wc -l
