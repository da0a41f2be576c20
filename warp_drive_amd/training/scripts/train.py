#!/usr/bin/env python3
"""End-to-end training on the HIP backend (mirror of the reference's
warp_drive/training/scripts/example_training_script_pycuda.py:41-225).

    python -m warp_drive_amd.training.scripts.train --env tag_continuous [--iters 5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        -m warp_drive_amd.training.scripts.train --env tag_continuous      # 8 x MI355X, RCCL DDP
"""
import argparse
import logging
import os

# kernel arguments in device memory instead of host-coherent memory (bench.py: one TagContinuous tick 54.3 -> 50.5 us);
# read when HIP initialises, so it is set before torch is imported
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import torch  # noqa: E402
import yaml

from warp_drive_amd import distributed as wdd
from warp_drive_amd.env_wrapper import EnvWrapper
from warp_drive_amd.envs.cartpole import CUDAClassicControlCartPoleEnv
from warp_drive_amd.envs.tag_continuous import TagContinuous
from warp_drive_amd.envs.tag_gridworld import CUDATagGridWorld
from warp_drive_amd.training.trainer import Trainer

_CONFIGS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "run_configs")
_ENVS = {"tag_continuous": TagContinuous, "tag_gridworld": CUDATagGridWorld,
         "single_cartpole": CUDAClassicControlCartPoleEnv}


def policy_map_for(name, env):
    if name == "tag_continuous":  # agent_type 1 = tagger
        return {"tagger": sorted(env.taggers), "runner": sorted(env.runners)}
    if name == "tag_gridworld":  # the last agent is the runner
        return {"tagger": list(range(env.num_taggers)), "runner": [env.num_agents - 1]}
    return {"shared": list(range(env.num_agents))}


def setup_trainer(env_name, overrides=None, results_dir=None, verbose=True):
    config = yaml.safe_load(open(os.path.join(_CONFIGS, f"{env_name}.yaml")))
    for section, kv in (overrides or {}).items():
        config.setdefault(section, {}).update(kv)
    rank, local_rank, world = wdd.rank_info()
    device = wdd.device_index(local_rank)  # one rank per GPU: device = local rank
    torch.cuda.set_device(device)
    wdd.pin_rank_to_cpus(local_rank, device=device)  # several ranks per node: each on its own CPUs, next to its GPU
    wdd.init_process_group(backend="nccl", device_id=device)
    env_cfg = dict(config["env"])
    # (identical seeded start state on every rank; the action streams differ by seed + rank)
    env = _ENVS[env_name](**env_cfg)
    wrapper = EnvWrapper(env_obj=env, num_envs=int(config["trainer"]["num_envs"]), env_backend="hip",
                         process_id=device)
    return Trainer(env_wrapper=wrapper, config=config, policy_tag_to_agent_id_map=policy_map_for(env_name, env),
                   device_id=device, results_dir=results_dir, verbose=verbose)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", "-e", choices=sorted(_ENVS), default="tag_continuous")
    ap.add_argument("--iters", type=int, default=None, help="training iterations (default: from num_episodes)")
    ap.add_argument("--num_envs", type=int, default=None)
    ap.add_argument("--train_batch_size", type=int, default=None)
    ap.add_argument("--results_dir", default=None)
    args = ap.parse_args()
    logging.getLogger().setLevel(logging.WARNING)
    overrides = {"trainer": {k: v for k, v in (("num_envs", args.num_envs),
                                               ("train_batch_size", args.train_batch_size)) if v is not None}}
    trainer = setup_trainer(args.env, overrides, args.results_dir)
    trainer.train(args.iters)
    trainer.graceful_close()
    if trainer.rank == 0:
        print(trainer.perf_stats.get_perf_stats())
    wdd.shutdown()


if __name__ == "__main__":
    main()
