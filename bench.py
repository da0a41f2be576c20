#!/usr/bin/env python3
"""bench.py -- env steps/sec of the TagContinuous rollout hot path on MI355X.

Workload (BASELINE.json configs[2]): TagContinuous, 5 taggers x 100 runners, partial
observation K = 10, run_configs/tag_continuous.yaml physics, num_envs = 2000 PER GPU.
One "step" = one rollout tick over all replicas of a rank = ONE launch of the fused tick kernel
HipTagContinuousTick_K10: sample both action heads (uniform synthetic policy output resident
in HBM) -> step -> reset finished replicas in place (`--unfused`: the same tick as four
launches: sample_actions x2, HipTagContinuousStep_K10, reset_when_done_fused),
replayed from C through the C-ABI (include/wd_hip.h) on the stream torch uses.
value = total env-steps of all ranks / max-over-ranks wall time of the timed region.
Replicas shard trivially (weak scaling): no collective in the data path; for N > 1 the
only RCCL traffic is the barrier + the max-reduction of the timing.

The cost of a TagContinuous tick follows the number of agents still in the game, which falls along an
episode (all replicas of a run restart together, as in WarpDrive training: 105 agents at tick 0, ~27 at
tick 500 under the uniform random policy).  The timed region therefore sits at a defined place of the
episode: it starts on an episode boundary when K covers at least one episode (the default K = 2000 is
four whole episodes, i.e. the episode average), otherwise it is centred on the middle of an episode,
whose tick costs what the episode costs on average (config.episode_window says where; the `episode`
object carries the whole profile measured in the same run).

The timed region holds the K launches and nothing else (no event records).  The dominant kernel's
average launch duration (roofline.achieved) is measured with HIP events on the launch stream around
windows of 25 back-to-back launches, over one whole episode just before and one just after the timed
region.  `roofline.frac` prices SURVEY's algorithmic bytes; `roofline.frac_moved` the HBM bytes the PMC counters
saw (`traffic`), `roofline.frac_full_load` the algorithmic bytes at the cost of a tick with every agent in the
game (the window of the episode profile that starts at the top of an episode).  `--no-tags` sets the tagging
distance to 0: nobody ever leaves the game, every tick is a full-load tick (the policy-independent worst case).
`ms_per_step_spread` = min / median / max of five repeats of the timed region at the same place of the
episode (`value` is the first).

Usage:  python bench.py [--gpus N] [--steps K] [--warmup W]
        (N > 1 without a launcher: bench.py starts its own N ranks through torch.distributed.run)
        python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

# kernel arguments in device memory instead of host-coherent memory (measured on MI355X: one
# TagContinuous tick 54.3 -> 50.5 us, TagGridWorld 10.9 -> 7.9 us); must be set before HIP initialises
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

# Shader cycles per wave-instruction per SIMD with four wavefronts resident (experiments/ubench,
# profiles/r02_ubench.txt), by the PMC counter classes of scripts/pmc_mix_tc.sh: v_add/mul_f32 ~1.2,
# v_fma_f32 ~2, integer / min / max / med3 / compare ~2.9 (the INT32 class of the tick is dominated by
# v_med3_u32), conversions ~2.9, float64 and 64-bit integer ~4, transcendental ~6, everything else
# (moves, selects, lane ops) ~2.5
VALU_CYCLES = {"ADD_F32": 1.2, "MUL_F32": 1.2, "FMA_F32": 2.0, "INT32": 2.9, "CVT": 2.9, "INT64": 4.0,
               "ADD_F64": 4.0, "MUL_F64": 4.0, "FMA_F64": 4.0, "TRANS_F32": 6.0, "OTHER": 2.5}
SIMDS = 256 * 4

BENCH_CFG = dict(num_taggers=5, num_runners=100, grid_length=20.0, episode_length=500, max_acceleration=0.1,
                 min_acceleration=-0.1, max_turn=2.356, min_turn=-2.356, num_acceleration_levels=20,
                 num_turn_levels=20, skill_level_runner=1.0, skill_level_tagger=1.0, max_speed=1.0, seed=274880,
                 use_full_observation=False, num_other_agents_observed=10, tagging_distance=0.02,
                 tag_reward_for_tagger=10.0, tag_penalty_for_runner=-10.0, step_penalty_for_tagger=-0.0,
                 step_reward_for_runner=0.0, edge_hit_penalty=-0.0, end_of_game_reward_for_runner=1.0,
                 runner_exits_game_after_tagged=True)


def step_algorithmic_bytes(N, K, full_obs):
    """bytes one TagContinuous env-step must move (4-byte elements):
    reads 5 state + still_in_game + 2 actions per agent + 12; writes 5 state + still_in_game
    + edge penalty + reward + F obs + K nearest ids per agent + 12."""
    F = 7 * (N - 1) + 1 if full_obs else 7 * K + 1
    k_ids = 0 if full_obs else K
    return 4 * N * (5 + 1 + 2) + 4 * N * (5 + 1 + 1 + 1 + F + k_ids) + 24


def valu_roofline(kernel, num_envs, full_obs, kernel_us):
    """Second roofline ("report both, grade on HBM"): the VALU issue time of one launch.
    Dynamic instruction counts per class come from the PMC passes of scripts/pmc_mix_tc.sh
    (profiles/pmc_mix.json, averaged over whole episodes), quoted only when they were collected on exactly
    the code object loaded now; cycles per instruction class from experiments/ubench (VALU_CYCLES)."""
    path = os.path.join(ROOT, "profiles", "pmc_mix.json")
    try:
        from warp_drive_amd.managers import hip_driver

        sha = hip_driver.code_object_sha256(kernel)
        rec = json.load(open(path))
        if (rec.get("hsaco_sha256") != sha or rec.get("kernel") != kernel or rec.get("num_envs") != num_envs
                or bool(rec.get("full_obs")) != full_obs):
            return None
        c = rec["counters_per_launch"]
        classes = {k: float(c.get("SQ_INSTS_VALU_" + k, 0.0)) for k in VALU_CYCLES if k != "OTHER"}
        classes["OTHER"] = max(0.0, float(c["SQ_INSTS_VALU"]) - sum(classes.values()))
        cycles = sum(classes[k] * VALU_CYCLES[k] for k in classes)  # summed over all wavefronts of a launch
        clock_ghz = float(rec.get("shader_clock_ghz", 2.3))
        issue_us = cycles / SIMDS / (clock_ghz * 1e3)
        return {"bound": "valu issue", "wave_insts": float(c["SQ_INSTS_VALU"]) / float(c["SQ_WAVES"]),
                "waves": float(c["SQ_WAVES"]), "insts_by_class_per_wave": {k: v / float(c["SQ_WAVES"]) for k, v in classes.items()},
                "cycles_per_inst_by_class": VALU_CYCLES, "shader_clock_ghz": clock_ghz,
                "issue_bound_us": issue_us, "frac": issue_us / kernel_us if kernel_us > 0 else None,
                "active_lanes_per_inst": float(c.get("SQ_THREAD_CYCLES_VALU", 0.0)) / max(float(c["SQ_INSTS_VALU"]), 1.0),
                "source": "profiles/pmc_mix.json"}
    except Exception:
        return None


def _cpu_quota_cores():
    """CPU bandwidth limit of this cgroup in cores (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited"""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if quota == "max" else float(quota) / float(period)
    except Exception:
        pass
    try:
        quota = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if quota <= 0 else quota / period
    except Exception:
        return None


def cpu_baseline(cfg, target_seconds=12.0):
    """The oracle's C restatement (oracle/csrc/wd_oracle.c) timed on this host's cores, on a bounded sample of the
    same workload: once on ONE core and once on every core this process may run on (its affinity mask, capped by the
    cgroup's CPU quota).  All ticks of a run happen inside one C call (replicas are independent: a thread takes chunks
    of replicas through all their ticks, dynamic schedule, no per-tick Python or numpy work).  Reported, never the
    optimisation target."""
    from oracle.tag_continuous_c import TagContinuousCOracle

    logical = os.cpu_count() or 1
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = logical
    quota = _cpu_quota_cores()
    cores = max(1, min(affinity, int(quota + 0.5)) if quota else affinity)
    rng = np.random.RandomState(0)

    def run(E, threads, seconds):
        orc = TagContinuousCOracle(E, n_threads=threads, **cfg)  # seeded start state (test infrastructure)
        na, nt = len(orc.acceleration_actions), len(orc.turn_actions)
        acts = np.stack([rng.randint(0, na, size=(E, orc.N)), rng.randint(0, nt, size=(E, orc.N))], 2).astype(np.int32)
        t0 = time.perf_counter()
        orc.run_ticks(acts, 2)  # warm-up (thread pool, page faults) + calibration
        per_tick = max((time.perf_counter() - t0) / 2, 1e-6)
        ticks = int(max(3, min(orc.T - 4, seconds / per_tick)))
        t0 = time.perf_counter()
        orc.run_ticks(acts, ticks)
        dt = time.perf_counter() - t0
        return E * ticks / dt, ticks, dt

    single, ticks1, dt1 = run(16, 1, 0.25 * target_seconds)
    # replicas per core sized so that a whole-episode run of the multi-core leg lasts about 0.75 x target
    E = cores * max(16, int(0.75 * target_seconds * single / 496))
    value, ticks, dt = (single, ticks1, dt1) if cores == 1 else run(E, cores, 0.75 * target_seconds)
    return {"value": value, "unit": "env_steps/s", "cores": cores, "kind": "port",
            "value_single_core": single, "cores_effective": value / single, "parallel_efficiency": value / single / cores,
            "host": {"logical_cpus": logical, "affinity_cpus": affinity, "cgroup_cpu_quota": quota},
            "sample": f"{E} replicas x {ticks} ticks of the same TagContinuous 5x100 K=10 step from the start of an "
                      f"episode, no resets (C restatement of the reference CPU step; {cores} OpenMP threads, chunks of 4 "
                      f"replicas scheduled dynamically, {dt:.1f} s; single core: 16 replicas x {ticks1} ticks, {dt1:.1f} s). "
                      "Orientation: the reference's own Python step() does ~49 env-steps/s per core (BASELINE.md section 3)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--num-envs", type=int, default=None, help="replicas per GPU (default: the BASELINE config's)")
    ap.add_argument("--workload", choices=("tag_continuous", "tag_gridworld", "cartpole"), default="tag_continuous",
                    help="tag_continuous = BASELINE configs[2] (the headline metric); the other two are the "
                         "configs[1] / configs[4] side workloads quoted in DESIGN.md")
    ap.add_argument("--full-obs", action="store_true", help="use_full_observation=True variant (F = 729)")
    ap.add_argument("--num-runners", type=int, default=None, help="TagContinuous: runners per replica (default 100)")
    ap.add_argument("--num-taggers", type=int, default=None, help="TagContinuous: taggers per replica (default 5)")
    ap.add_argument("--mode", choices=("plan", "graph"), default="plan",
                    help="plan: launches replayed from C; graph: hipGraph of 10 ticks")
    ap.add_argument("--no-tags", action="store_true",
                    help="TagContinuous: tagging_distance = 0, so no runner is ever tagged and all agents stay in the "
                         "game for the whole episode: the worst case of the tick, independent of the policy")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reset", action="store_true", help="leave reset_when_done_fused out of the tick")
    ap.add_argument("--ticks-per-launch", type=int, default=1,
                    help="Cartpole / TagGridWorld: env ticks fused into one launch, every tick recorded in batch tensors "
                         "(fixed-policy rollout; SURVEY 8(d) asks the ceiling run to fuse T ticks); a bench 'step' is "
                         "then one launch = T ticks")
    ap.add_argument("--no-spread", action="store_true", help="skip the four extra repeats of the timed region")
    ap.add_argument("--profile-episodes", type=int, default=0,
                    help="profiler helper: run this many whole episodes of ticks and exit (no timing, no JSON)")
    ap.add_argument("--trainer-leg", choices=("auto", "on", "off"), default="auto",
                    help="add a `trainer` object to the line: ONE timed PPO training iteration of BASELINE configs[3] per rank "
                         "(2000 replicas x 250 ticks, both policies, the real one-bucket gradient all-reduce) after one warm-up "
                         "iteration -- rollout_ms, update_ms, allreduce_us measured inside the iteration, end-to-end env-steps/s, "
                         "collectives per iteration, parameter checksum per rank.  auto = when --gpus > 1 (tag_continuous only)")
    ap.add_argument("--trainer-timeout", type=float, default=300.0,
                    help="seconds after which the trainer leg is abandoned (the kernel line is printed without it)")
    ap.add_argument("--trainer-num-envs", type=int, default=2000, help="replicas per rank of the trainer leg")
    ap.add_argument("--trainer-ticks", type=int, default=250, help="ticks per training iteration of the trainer leg")
    ap.add_argument("--unfused", action="store_true",
                    help="tick = 4 launches (sample x2, step, fused reset) instead of the single tick kernel")
    args = ap.parse_args()

    from warp_drive_amd import distributed as wdd

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as a plain `python bench.py --gpus N`: become the launcher of N ranks (one per GPU,
        # localhost rendezvous), which print the JSON line themselves
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "4")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))

    rank, local_rank, world = wdd.rank_info()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU fallback for the HIP path"
    device = wdd.device_index(local_rank)
    torch.cuda.set_device(device)
    pinned = wdd.pin_rank_to_cpus(local_rank, device=device)  # N > 1: every rank on its own CPUs, next to its GPU
    wdd.init_process_group(backend="nccl", device_id=device)  # RCCL; no-op for one rank

    from warp_drive_amd.env_wrapper import EnvWrapper
    from warp_drive_amd.envs.tag_continuous import TagContinuous
    from warp_drive_amd.managers.function_manager import HIPSampler
    from warp_drive_amd.rollout import RolloutEngine
    from warp_drive_amd.training.data_loader import create_and_push_data_placeholders

    from warp_drive_amd.envs.cartpole import CUDAClassicControlCartPoleEnv
    from warp_drive_amd.envs.tag_gridworld import CUDATagGridWorld

    if args.workload == "tag_continuous":
        cfg = dict(BENCH_CFG, use_full_observation=bool(args.full_obs))
        if args.num_runners is not None:
            cfg["num_runners"] = args.num_runners
        if args.num_taggers is not None:
            cfg["num_taggers"] = args.num_taggers
        if args.no_tags:
            cfg["tagging_distance"] = 0.0
        E = args.num_envs or 2000
        env_obj = TagContinuous(**cfg)
    elif args.workload == "tag_gridworld":  # BASELINE configs[1]
        cfg = dict(num_taggers=4, grid_length=10, episode_length=100, seed=27, wall_hit_penalty=0.1,
                   tag_reward_for_tagger=10.0, tag_penalty_for_runner=2.0, step_cost_for_tagger=0.01,
                   use_full_observation=True)
        E = args.num_envs or 1000
        env_obj = CUDATagGridWorld(**cfg)
    else:  # BASELINE configs[4]
        cfg = dict(episode_length=500, seed=274880)
        E = args.num_envs or 100000
        env_obj = CUDAClassicControlCartPoleEnv(**cfg)
        env_obj.ticks_per_launch = max(1, args.ticks_per_launch)
    w = EnvWrapper(env_obj=env_obj, num_envs=E, env_backend="hip", process_id=device)
    w.reset_all_envs()
    sampler = HIPSampler(w.cuda_function_manager)
    seed = wdd.rank_seed(cfg["seed"], rank)  # seed + rank, trainer_base.py:249-252
    sampler.init_random(seed=seed)
    seeds = wdd.gather_ints(seed)
    create_and_push_data_placeholders(env_wrapper=w, action_sampler=sampler, training_batch_size_per_env=None,
                                      push_data_batch_placeholders=False)
    rollout_batch = None
    if args.workload == "tag_gridworld" and args.ticks_per_launch > 1:
        Tn, dev, Ng = args.ticks_per_launch, torch.device("cuda", device), env_obj.num_agents
        env_obj.ticks_per_launch = Tn
        rollout_batch = {"obs": torch.zeros((Tn, E, Ng, 4 * Ng + 1), dtype=torch.float32, device=dev),
                         "actions": torch.zeros((Tn, E, Ng, 1), dtype=torch.int32, device=dev),
                         "rewards": torch.zeros((Tn, E, Ng), dtype=torch.float32, device=dev),
                         "done": torch.zeros((Tn, E), dtype=torch.int32, device=dev)}
    if args.workload == "cartpole" and args.ticks_per_launch > 1:
        # the ceiling run: T ticks per launch, every tick recorded in the trainer's [T, E, ...] batch tensors
        # (distinct addresses per tick: T ticks move T times the bytes)
        Tn, dev = args.ticks_per_launch, torch.device("cuda", device)
        rollout_batch = {"obs": torch.zeros((Tn, E, 1, 4), dtype=torch.float32, device=dev),
                         "actions": torch.zeros((Tn, E, 1, 1), dtype=torch.int32, device=dev),
                         "rewards": torch.zeros((Tn, E, 1), dtype=torch.float32, device=dev),
                         "done": torch.zeros((Tn, E), dtype=torch.int32, device=dev)}
    engine = RolloutEngine(w, sampler, probabilities=None, reset_done=not args.no_reset, fused=not args.unfused,
                           rollout_batch=rollout_batch)
    steps, warmup = args.steps, args.warmup
    if args.mode == "graph":
        steps = max(10, steps // 10 * 10)
        warmup = max(10, warmup // 10 * 10)

    def run(n):
        if n <= 0:
            return
        engine.run(n) if args.mode == "plan" else engine.run_graph(n, 10)

    barrier = wdd.barrier
    T = int(cfg["episode_length"])
    is_tc = args.workload == "tag_continuous"
    if args.profile_episodes:
        # profiler runs (scripts/collect_profiles.sh, scripts/pmc_mix_tc.sh): whole episodes and nothing
        # else, so that a per-dispatch average over the run is the episode average
        engine.run(args.profile_episodes * T)
        torch.cuda.synchronize()
        wdd.shutdown()
        return

    # hipGraph capture is not allowed on the legacy default stream: use a side stream there
    side = torch.cuda.Stream() if args.mode == "graph" else None
    if side is not None:
        torch.cuda.set_stream(side)
    ticks_done = 0  # launches since the reset (= the episode tick of every replica, mod T)
    run(warmup)
    ticks_done += warmup
    barrier()

    WIN = 25

    def time_episode(launches):
        """per-window average launch duration (us) of the dominant kernel over `launches` back-to-back
        launches: HIP events on the launch stream around windows of WIN launches (outside the timed region)"""
        if len(engine.entry_names) > 1:  # unfused tick: the plan times its step entry itself
            engine.plan.enable_timing(engine.step_entry, 8, max(1, launches // 8))
            engine.run(launches)
            torch.cuda.synchronize()
            ms, n = engine.plan.read_timing()
            engine.plan.enable_timing(-1, 1, 1)
            return [ms / max(n, 1) * 1e3] * max(1, launches // WIN)
        evs = []
        for _ in range(launches // WIN):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            engine.run(WIN)
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        return [e0.elapsed_time(e1) * 1e3 / WIN for e0, e1 in evs]

    P = (max(T, 500) + WIN - 1) // WIN * WIN if args.mode == "plan" else 500
    first_pass_tick = ticks_done % T
    win_us = time_episode(P)   # one whole episode; also leaves the clocks where a long run has them
    ticks_done += P
    # the timed window: on an episode boundary when it covers whole episodes, else centred mid-episode
    t0 = 0 if steps >= T else (T - steps) // 2
    if is_tc and args.mode == "plan":
        pre = (t0 - ticks_done) % T
        run(pre)
        ticks_done += pre

    def timed_region():
        barrier()
        t_start = time.perf_counter()
        run(steps)                    # the timed region: K launches, no event records
        torch.cuda.synchronize()
        dt_local = time.perf_counter() - t_start   # this rank's own time (before it waits for the others)
        barrier()
        return wdd.max_over_ranks(time.perf_counter() - t_start), dt_local

    window_first_tick = ticks_done % T
    live_first = None
    if is_tc:
        torch.cuda.synchronize()
        live_first = float(w.cuda_data_manager.pull_data_from_device("still_in_the_game").sum(axis=1).mean())
    elapsed, elapsed_local = timed_region()
    ticks_done += steps
    repeats = [elapsed]
    for _ in range(0 if args.no_spread else 4):  # the same K launches at the same place of the episode
        pre = (window_first_tick - ticks_done) % T if is_tc else 0
        run(pre)
        ticks_done += pre
        repeats.append(timed_region()[0])
        ticks_done += steps
    per_rank_ms = wdd.gather_floats(elapsed_local / steps * 1e3)
    allreduce_us = wdd.time_allreduce(190550, 50) if world > 1 else None  # both policies' gradients, one bucket
    second_pass_tick = ticks_done % T
    win_us2 = time_episode(P)
    kern_us = (sum(win_us) + sum(win_us2)) / (len(win_us) + len(win_us2))
    kern_n = (len(win_us) + len(win_us2)) * WIN

    out = None
    if rank == 0:
        N = w.n_agents
        if args.workload == "tag_continuous":
            K = cfg["num_other_agents_observed"]
            bytes_per_env_step = step_algorithmic_bytes(N, K, cfg["use_full_observation"])
            shape = f"{cfg['num_taggers']} taggers x {cfg['num_runners']} runners"
            label = (("BASELINE configs[2]: " if N == 105 else "") + ("NO TAGS (tagging_distance 0: every agent "
                     "stays in the game) " if args.no_tags else "") + f"TagContinuous {shape}, "
                     f"{f'full obs F={7 * (N - 1) + 1}' if args.full_obs else 'partial obs K=10 (F=71)'}")
            metric = f"env steps/sec, TagContinuous {shape}"
        elif args.workload == "tag_gridworld":
            bytes_per_env_step = 576  # N=5, full obs F=21
            label, metric = "BASELINE configs[1]: TagGridWorld 10x10, 5 agents, full obs", "env steps/sec, TagGridWorld"
        else:
            bytes_per_env_step = 68
            label, metric = "BASELINE configs[4]: Cartpole-v1 Euler step, 1 agent", "env steps/sec, Cartpole"
        if engine.fused:
            # a fused tick kernel also reads every head's probabilities and reads+writes the RNG epoch
            bytes_per_env_step += sum(4 * N * a for a in engine.head_sizes) + 8 * N
        bytes_per_launch = bytes_per_env_step * E
        if rollout_batch is not None and args.workload == "cartpole":
            # T ticks per launch with every tick recorded: per env-step 16 B observation + 4 B action + 4 B reward +
            # 4 B done flag written (28 B, distinct addresses), per launch and replica the state in and out, the
            # probabilities, the RNG epoch and the time step (SURVEY 8(d)'s 68 B counted once)
            bytes_per_launch = (28 * engine.ticks_per_launch + bytes_per_env_step) * E
        elif rollout_batch is not None:
            # TagGridWorld: per env-step N x (F + 2) floats recorded + the done flag; the per-tick arrays once per launch
            bytes_per_launch = ((4 * N * (4 * N + 1 + 2) + 4) * engine.ticks_per_launch + bytes_per_env_step) * E
        kern_s = kern_us * 1e-6
        achieved = bytes_per_launch / kern_s / 1e9 if kern_s > 0 else 0.0
        # HBM bytes per launch from the PMC passes (scripts/collect_profiles.sh): only quoted when they
        # were collected on exactly the code object loaded now and at this shape, else null
        traffic, traffic_stale = None, False
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                from warp_drive_amd.managers import hip_driver

                sha = hip_driver.code_object_sha256(engine.step_kernel_name)
                shape_recs = [r for r in json.load(open(pmc)).values()
                              if r.get("kernel") == engine.step_kernel_name and r.get("num_envs") == E
                              and r.get("full_obs") == bool(args.full_obs) and N == 105
                              and bool(r.get("no_tags", False)) == bool(args.no_tags)]
                recs = [r for r in shape_recs if r.get("hsaco_sha256") == sha]
                if args.workload == "tag_continuous" and recs:
                    traffic = recs[0].get("hbm_bytes_per_launch")
                # counters exist for this shape but were collected on another build of the kernels
                traffic_stale = bool(args.workload == "tag_continuous" and shape_recs and not recs)
            except Exception:
                traffic = None
        # the tick with every agent in the game: the window of the episode profile closest to the top of an episode
        full_load_us = full_load_tick = None
        if is_tc and args.mode == "plan":
            starts = [((first_pass_tick + i * WIN) % T, v) for i, v in enumerate(win_us)] + \
                     [((second_pass_tick + i * WIN) % T, v) for i, v in enumerate(win_us2)]
            top = [sv for sv in starts if sv[0] < WIN]
            if top:
                full_load_tick = min(sv[0] for sv in top)
                vals = [v for st, v in top if st == full_load_tick]
                full_load_us = sum(vals) / len(vals)
        out = {
            "metric": metric,
            "value": world * E * steps * engine.ticks_per_launch / elapsed,
            "unit": "env_steps/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{label}, num_envs={E} per GPU; tick = sample_actions ({len(engine.head_sizes)} head"
                            f"{'s' if len(engine.head_sizes) > 1 else ''}) + step"
                            f"{'' if args.no_reset else ' + reset of finished replicas'}"
                            f"{' (one fused launch)' if engine.fused else ''}",
                "num_envs_per_gpu": E, "num_agents": N, "launch_mode": args.mode,
                "kernels_per_tick": len(engine.entry_names), "ticks_per_launch": engine.ticks_per_launch,
                "parallelism": f"env-replica sharding x{world}", "sampler_seeds": seeds,
                "episode_window": {"first_tick": window_first_tick, "last_tick": (window_first_tick + steps - 1) % T,
                                   "episode_length": T, "whole_episodes": steps // T,
                                   "agents_in_the_game_at_first_tick": live_first},
            },
            "ms_per_step_spread": {"min": min(repeats) / steps * 1e3, "median": sorted(repeats)[len(repeats) // 2] / steps * 1e3,
                                   "max": max(repeats) / steps * 1e3, "repeats": len(repeats)},
            "per_rank_ms_per_step": per_rank_ms,
            "cpus_of_rank0": (len(pinned) if pinned else None),  # N > 1: CPUs this rank was pinned to (next to its GPU)
            "allreduce_us": allreduce_us,
            "roofline": {
                "bound": "hbm", "kernel": engine.step_kernel_name, "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_stale": traffic_stale,
                # the same launch duration priced on the bytes the counters saw (rows of agents out of the game are
                # not rewritten: less than the algorithmic bytes) ...
                "frac_moved": (traffic / kern_s / 1e9 / HBM_PEAK_GBS) if (traffic and kern_s > 0) else None,
                # ... and the algorithmic bytes at the cost of a tick with every agent in the game
                "frac_full_load": (bytes_per_launch / (full_load_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if full_load_us else None,
                "full_load_us": full_load_us, "full_load_window_first_tick": full_load_tick,
                "algorithmic_bytes_per_launch": bytes_per_launch, "avg_kernel_us": kern_s * 1e6,
                "samples": kern_n,
                "timing": f"HIP events around windows of {WIN} back-to-back launches, one whole episode "
                          f"({P} launches) before + one after the timed region",
            },
            # tick cost along the episode, from the two timing passes (all replicas are at the same tick)
            "episode": {"window_ticks": WIN, "first_pass_starts_at_tick": first_pass_tick,
                        "second_pass_starts_at_tick": second_pass_tick,
                        "us_per_tick_mean": kern_us, "us_per_tick_min": min(win_us + win_us2),
                        "us_per_tick_max": max(win_us + win_us2),
                        "us_per_tick_windows_first_pass": [round(v, 2) for v in win_us]},
            "roofline_valu": valu_roofline(engine.step_kernel_name, E, bool(args.full_obs), kern_us)
            if is_tc else None,
        }
        if is_tc and args.mode == "plan" and kern_us > 0:
            # The cost of a TagContinuous tick falls along the episode (agents leave the game), so a `value` timed over a
            # window that is not a whole number of episodes is the rate of THAT window (the contract: exactly K timed
            # steps), not of the workload.  The rate that belongs next to `roofline.frac` -- both from the two
            # whole-episode HIP-event passes -- is always printed beside it, and the line says which kind `value` is.
            whole = (steps % T == 0) and steps >= T
            out["value_episode_average"] = world * E / (kern_us * 1e-6)
            out["value_window"] = {"whole_episodes": whole, "first_tick": window_first_tick, "ticks": steps,
                                   "note": None if whole else "`value` is the rate of a window shorter than (or not a multiple "
                                   "of) an episode; quote `value_episode_average` next to roofline.frac"}
        if not args.no_cpu_baseline and args.workload == "tag_continuous" and world == 1:  # N = 1 only
            try:
                out["cpu_baseline"] = cpu_baseline({k: v for k, v in cfg.items()})
            except Exception as err:  # the baseline is reported context, never a reason to lose the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "env_steps/s", "cores": os.cpu_count(),
                                       "kind": "port", "sample": f"failed: {err}"}

    # BASELINE configs[3] is a TRAINING configuration (PPO + the gradient all-reduce over RCCL): at N > 1 the line also
    # carries one training iteration timed from the inside, on every rank (collectives inside: all ranks take part).
    # The kernel line above is the contract and is complete by now: whatever the trainer leg does -- an exception (reported
    # in the line; the ranks agree on it before they enter a collective, training/bench_iteration.py), or a rank that dies
    # or hangs inside a collective (nobody can be called back from there) -- the line is printed and the job ends: a
    # watchdog on every rank prints it (rank 0) and leaves after --trainer-timeout seconds.
    if is_tc and (args.trainer_leg == "on" or (args.trainer_leg == "auto" and world > 1)):
        import threading

        from warp_drive_amd.training.bench_iteration import run_configs3_iteration

        def give_up():
            if rank == 0:
                out["trainer"] = {"failed": f"no result after {args.trainer_timeout} s: a rank failed or a collective hung; "
                                            "the kernel line above is unaffected"}
                print(json.dumps(out), flush=True)
            sys.stderr.write(f"bench.py: rank {rank}: trainer leg abandoned after {args.trainer_timeout} s\n")
            sys.stderr.flush()
            os._exit(0)

        watchdog = threading.Timer(float(args.trainer_timeout), give_up)
        watchdog.daemon = True
        watchdog.start()
        try:
            trainer_leg = run_configs3_iteration(args.trainer_num_envs, args.trainer_ticks, warmup_iterations=1)
            trainer_leg["hardware_note"] = ("RCCL over xGMI" if world > 1 and os.environ.get("WD_DIST_BACKEND") in (None, "", "nccl")
                                            else "single rank: no collective" if world == 1 else
                                            f"backend {os.environ.get('WD_DIST_BACKEND')} (not RCCL)")
        except Exception as err:  # the kernel line is the contract; a failed trainer leg must not lose it
            import traceback

            traceback.print_exc()
            trainer_leg = {"failed": f"{type(err).__name__}: {err}"}
        watchdog.cancel()
        if rank == 0:
            out["trainer"] = trainer_leg
    if rank == 0:
        print(json.dumps(out), flush=True)
    wdd.shutdown()


if __name__ == "__main__":
    main()
