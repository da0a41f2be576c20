"""Trainer rollout for Cartpole at configs[4] (100 000 replicas, 10-tick batches = 1e6 env-steps per iteration):
the per-tick path (policy forward + fused env tick + bookkeeping per tick) vs the whole batch as ONE launch
with the policy evaluated inside the env kernel.  Run on the GPU box."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from warp_drive_amd.training.scripts.train import setup_trainer
for one_launch, batch in ((False, 1000000), (True, 1000000), (True, 5000000)):
    ov = {"trainer": {"fused_rollout_policy": one_launch, "train_batch_size": batch, "num_episodes": 10000000}}
    tr = setup_trainer("single_cartpole", ov, results_dir=f"/tmp/cp_{int(one_launch)}_{batch}", verbose=False)
    tr._generate_rollout_batch(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): tr._generate_rollout_batch()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    tr.train(2); torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train(3)
    torch.cuda.synchronize()
    it = (time.perf_counter() - t0) / 3
    print(f"whole batch in one launch={one_launch}, kernel {tr.engine.step_kernel_name}: rollout of {tr.batch_len} ticks x {tr.num_envs} replicas = "
          f"{dt*1e3:.2f} ms -> {tr.train_batch_size/dt:.3e} env-steps/s; training iteration {it*1e3:.1f} ms -> {tr.train_batch_size/it:.3e} env-steps/s end to end")
    tr.graceful_close()
