#!/usr/bin/env python3
"""Tick time of the fused TagContinuous tick along an episode (BASELINE configs[2], 2000 replicas).

Every replica of a launch is at the same episode tick (fixed-length episodes: the replicas of a WarpDrive
run restart together), so the cost of a tick follows the number of agents still in the game.  Prints, per
window of `--window` ticks: us per tick (HIP events on the launch stream) and the mean number of agents in
the game at the end of the window.  Run on the GPU box:  python scripts/episode_profile.py [--episodes 2]
"""
import argparse
import json
import os
import sys

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-envs", type=int, default=2000)
    ap.add_argument("--window", type=int, default=25)
    ap.add_argument("--episodes", type=int, default=2)
    args = ap.parse_args()
    from warp_drive_amd.env_wrapper import EnvWrapper
    from warp_drive_amd.envs.tag_continuous import TagContinuous
    from warp_drive_amd.managers.function_manager import HIPSampler
    from warp_drive_amd.rollout import RolloutEngine
    from warp_drive_amd.training.data_loader import create_and_push_data_placeholders

    cfg = dict(bench.BENCH_CFG)
    T = cfg["episode_length"]
    w = EnvWrapper(env_obj=TagContinuous(**cfg), num_envs=args.num_envs, env_backend="hip")
    w.reset_all_envs()
    sampler = HIPSampler(w.cuda_function_manager)
    sampler.init_random(seed=cfg["seed"])
    create_and_push_data_placeholders(env_wrapper=w, action_sampler=sampler, training_batch_size_per_env=None,
                                      push_data_batch_placeholders=False)
    engine = RolloutEngine(w, sampler)
    engine.run(T)  # one whole episode: clocks up, replicas back at tick 0
    torch.cuda.synchronize()
    rows = []
    for ep in range(args.episodes):
        for t0 in range(0, T, args.window):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            engine.run(args.window)
            e1.record()
            torch.cuda.synchronize()
            live = float(w.cuda_data_manager.pull_data_from_device("still_in_the_game").sum(axis=1).mean())
            rows.append({"episode": ep, "t_end": t0 + args.window, "us_per_tick": e0.elapsed_time(e1) * 1e3 / args.window,
                         "live_agents": live})
            print(f"episode {ep} ticks {t0:3d}..{t0 + args.window:3d}: {rows[-1]['us_per_tick']:7.2f} us/tick, "
                  f"{live:6.1f} agents in the game", flush=True)
    mean = sum(r["us_per_tick"] for r in rows) / len(rows)
    print(json.dumps({"kernel": engine.step_kernel_name, "num_envs": args.num_envs, "mean_us_per_tick": mean,
                      "env_steps_per_s": args.num_envs / mean * 1e6, "windows": rows}))


if __name__ == "__main__":
    main()
