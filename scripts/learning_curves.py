#!/usr/bin/env python3
"""Learning curves of the trainer on the MI355X: "Mean episodic reward" per logged iteration for a few small, seeded
configurations (the numbers tests/test_gpu_learning.py's thresholds were chosen from; reference README.md:60 shows the
Cartpole curve of the reference trainer).

    python scripts/learning_curves.py [--which cartpole_kernel cartpole_tick gridworld_kernel gridworld_tick] [--iters N]
"""
import argparse
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import torch  # noqa: E402

CONFIGS = {
    # Cartpole pays 1 per tick: the mean episodic reward IS the mean episode length
    "cartpole_kernel": ("single_cartpole", {"trainer": {"num_envs": 1000, "train_batch_size": 1000 * 50, "num_episodes": 10 ** 6, "seed": 7},
                                            "env": {"episode_length": 200}}, "shared"),
    "cartpole_tick": ("single_cartpole", {"trainer": {"num_envs": 1000, "train_batch_size": 1000 * 50, "num_episodes": 10 ** 6, "seed": 7,
                                                      "fused_rollout_policy": False},
                                          "env": {"episode_length": 200}}, "shared"),
    "gridworld_kernel": ("tag_gridworld", {"trainer": {"num_envs": 600, "train_batch_size": 600 * 100, "num_episodes": 10 ** 6, "seed": 7},
                                           "policy": {p: {"to_train": True, "algorithm": "A2C", "vf_loss_coeff": 1, "entropy_coeff": 0.05,
                                                          "gamma": 0.98, "lr": 0.005, "model": {"type": "fully_connected", "fc_dims": [32, 32],
                                                                                               "model_ckpt_filepath": ""}}
                                                      for p in ("runner", "tagger")}}, "tagger"),
    "gridworld_tick": ("tag_gridworld", {"trainer": {"num_envs": 600, "train_batch_size": 600 * 100, "num_episodes": 10 ** 6, "seed": 7}}, "tagger"),
    # the taggers learn against a runner that stays the random initial policy: their episodic reward can only rise
    "gridworld_kernel_frozen_runner": ("tag_gridworld", {
        "trainer": {"num_envs": 600, "train_batch_size": 600 * 100, "num_episodes": 10 ** 6, "seed": 7},
        "policy": {p: {"to_train": p == "tagger", "algorithm": "A2C", "vf_loss_coeff": 1, "entropy_coeff": 0.05, "gamma": 0.98, "lr": 0.005,
                       "model": {"type": "fully_connected", "fc_dims": [32, 32], "model_ckpt_filepath": ""}} for p in ("runner", "tagger")}}, "tagger"),
    "gridworld_tick_frozen_runner": ("tag_gridworld", {
        "trainer": {"num_envs": 600, "train_batch_size": 600 * 100, "num_episodes": 10 ** 6, "seed": 7},
        "policy": {p: {"to_train": p == "tagger", "algorithm": "A2C", "vf_loss_coeff": 1, "entropy_coeff": 0.05, "gamma": 0.98, "lr": 0.002,
                       "model": {"type": "fully_connected", "fc_dims": [256, 256], "model_ckpt_filepath": ""}} for p in ("runner", "tagger")}}, "tagger"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--which", nargs="*", default=list(CONFIGS))
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--lr", type=float, default=None, help="override every policy's learning rate")
    ap.add_argument("--every", type=int, default=1, help="print every n-th iteration's value")
    ap.add_argument("--grid", type=int, default=None, help="TagGridWorld: grid_length override")
    ap.add_argument("--graph", default=None, help="trainer.graph_rollout override (true / false / auto)")
    args = ap.parse_args()
    from warp_drive_amd.training.scripts.train import setup_trainer

    for name in args.which:
        env, ov, pol = CONFIGS[name]
        ov = json.loads(json.dumps(ov))
        ov["saving"] = {"metrics_log_freq": 1, "model_params_save_freq": 0}
        if args.grid is not None and env == "tag_gridworld":
            ov.setdefault("env", {})["grid_length"] = args.grid
        if args.graph is not None:
            ov["trainer"]["graph_rollout"] = args.graph
        if args.lr is not None:
            import yaml

            base = yaml.safe_load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "warp_drive_amd",
                                                    "training", "run_configs", f"{env}.yaml")))
            ov.setdefault("policy", {})
            for p in base["policy"]:
                ov["policy"].setdefault(p, dict(base["policy"][p]))["lr"] = args.lr
        torch.manual_seed(args.seed)
        with tempfile.TemporaryDirectory() as d:
            tr = setup_trainer(env, ov, results_dir=d, verbose=False)
            t0 = time.time()
            tr.train(args.iters)
            dt = time.time() - t0
            curve = [json.loads(line)[pol]["Mean episodic reward"] for line in open(os.path.join(d, "results.json"))]
            tr.graceful_close()
        print(f"{name} (lr {args.lr}, graph {args.graph}, grid {args.grid}, seed {args.seed}): path={'one launch per batch' if tr._batch_rollout is not None else 'per tick'} "
              f"{args.iters} iterations in {dt:.1f} s; {pol} mean episodic reward per iteration:")
        print("   " + " ".join(f"{v:.2f}" for v in curve[:: args.every]), flush=True)


if __name__ == "__main__":
    main()
