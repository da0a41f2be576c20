"""Trainer rollout for TagGridWorld at configs[1] (1000 replicas, 100-tick batches = 100 000 env-steps per iteration,
two [256, 256] policies): policy forwards + fused env tick + bookkeeping per tick.  Run on the GPU box."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from warp_drive_amd.training.scripts.train import setup_trainer
for fused in (True, False):
    ov = {"trainer": {"fused_policy_forward": fused}}
    tr = setup_trainer("tag_gridworld", ov, results_dir=f"/tmp/gw_rt_{int(fused)}", verbose=False)
    tr._generate_rollout_batch(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): tr._generate_rollout_batch()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    tr.train(2); torch.cuda.synchronize()
    s0 = tr.perf_stats; r0, u0 = s0.rollout_time, s0.training_time
    t0 = time.perf_counter()
    tr.train(4)
    torch.cuda.synchronize()
    it = (time.perf_counter() - t0) / 4
    print(f"fused_policy_forward={fused}: rollout of {tr.batch_len} ticks x {tr.num_envs} replicas = {dt * 1e3:.2f} ms -> "
          f"{dt / tr.batch_len * 1e6:.1f} us/tick, {tr.batch_len * tr.num_envs / dt:.3e} env-steps/s; training iteration {it * 1e3:.1f} ms "
          f"-> {tr.batch_len * tr.num_envs / it:.3e} env-steps/s end to end", flush=True)
