"""Trainer rollout for TagGridWorld at configs[1] (1000 replicas, 100-tick batches = 100 000 env-steps per iteration,
"tagger" + "runner" policies): the shipped [256, 256] policies on the per-tick path (framework forward below
`fused_policy_forward_min_rows`, the fused forward kernel when forced), and [32, 32] / [64, 64] policies evaluated INSIDE
the env's rollout kernel (one launch per training batch) against the per-tick path.  Run on the GPU box."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from warp_drive_amd.training.scripts.train import setup_trainer


def small(h):
    pol = {"to_train": True, "algorithm": "A2C", "vf_loss_coeff": 1, "entropy_coeff": 0.05, "gamma": 0.98, "lr": 0.002,
           "model": {"type": "fully_connected", "fc_dims": [h, h], "model_ckpt_filepath": ""}}
    return {"runner": dict(pol), "tagger": dict(pol)}


CASES = [("[256,256] default (framework forward: 4000 + 1000 rows)", {}, {}),
         ("[256,256] default, tick replayed from a hipGraph", {"graph_rollout": True}, {}),
         ("[256,256] fused forward kernel forced", {"fused_policy_forward_min_rows": 0}, {}),
         ("[256,256] fused forward kernel forced, tick replayed from a hipGraph", {"fused_policy_forward_min_rows": 0, "graph_rollout": True}, {}),
         ("[32,32] policies inside the rollout kernel (one launch per batch)", {}, small(32)),
         ("[32,32] per-tick path", {"fused_rollout_policy": False}, small(32)),
         ("[64,64] policies inside the rollout kernel (one launch per batch)", {}, small(64)),
         ("[64,64] per-tick path", {"fused_rollout_policy": False}, small(64))]
for i, (label, trainer_ov, policy_ov) in enumerate(CASES):
    tr = setup_trainer("tag_gridworld", {"trainer": trainer_ov, "policy": policy_ov}, results_dir=f"/tmp/gw_rt_{i}", verbose=False)
    tr._generate_rollout_batch(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): tr._generate_rollout_batch()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): tr.engine.run(1)
    torch.cuda.synchronize()
    plan = (time.perf_counter() - t0) / 3
    tr.train(2); torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train(4)
    torch.cuda.synchronize()
    it = (time.perf_counter() - t0) / 4
    print(f"{label}: kernel {tr.engine.step_kernel_name} ({tr.engine.ticks_per_launch} ticks per launch, {plan * 1e6:.1f} us per launch); rollout of {tr.batch_len} ticks x {tr.num_envs} replicas = {dt * 1e3:.2f} ms -> "
          f"{dt / tr.batch_len * 1e6:.1f} us/tick, {tr.batch_len * tr.num_envs / dt:.3e} env-steps/s; training iteration {it * 1e3:.1f} ms "
          f"-> {tr.batch_len * tr.num_envs / it:.3e} env-steps/s end to end", flush=True)
    tr.graceful_close()
