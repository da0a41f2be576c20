"""Trainer at configs[2] (2000 replicas, 50-tick batches = 100 000 env-steps per iteration): the rollout
(policy forward + fused env tick + bookkeeping) and one whole training iteration, float32 (reference
semantics) vs bf16-autocast update.  Run on the GPU box; the output is kept as profiles/r0N_rollout_timing.txt."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from warp_drive_amd.training.scripts.train import setup_trainer
CASES = [("float32", "float32", True, True, "bf16x3"), ("float32", "float32", True, True, "bf16x3-no-reuse"),
         ("float32", "float32", True, True, "float32"),
         ("float32", "float32", True, False, "float32"), ("bfloat16", "float32", True, True, "bf16x3"),
         ("bfloat16", "bfloat16", False, False, "float32")]
if len(sys.argv) > 1:
    CASES = CASES[: int(sys.argv[1])]
for update, rollout, fused, fast, arith in CASES:
    ov = {"trainer": {"num_envs": 2000, "train_batch_size": 100000, "rollout_dtype": rollout, "update_dtype": update,
                      "fused_policy_forward": fused, "fused_tick": fast, "policy_arithmetic": arith.split("-")[0],
                      "reuse_rollout_activations": "no-reuse" not in arith}}
    tr = setup_trainer("tag_continuous", ov, results_dir=f"/tmp/rt_{update}_{rollout}_{int(fused)}_{int(fast)}_{arith}", verbose=False)
    for _ in range(3):  # (the second rollout of a fresh trainer carries a one-off ~70 ms: 22 / 91 / 20 / 20 ms measured)
        tr._generate_rollout_batch()
    torch.cuda.synchronize()
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
    print(f"update_dtype={update} rollout_dtype={rollout} fused_policy_forward={fused} fused_tick={tr._fast_tick is not None} policy_arithmetic={arith if fused else 'framework'}: rollout of {tr.batch_len} ticks = {dt*1e3:.1f} ms "
          f"-> {dt/tr.batch_len*1e3:.3f} ms/tick, {tr.train_batch_size/dt:.3e} env-steps/s; training iteration {it*1e3:.0f} ms "
          f"(rollout {(s0.rollout_time - r0)/4*1e3:.0f} + update {(s0.training_time - u0)/4*1e3:.0f}) -> {tr.train_batch_size/it:.3e} env-steps/s end to end", flush=True)
    tr.graceful_close()
