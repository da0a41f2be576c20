"""Trainer rollout at configs[2] (2000 replicas, 50-tick batches): eager vs hipGraph replay, float32 vs
bf16-autocast policy forward; and one training iteration end to end.  Run on the GPU box."""
import sys, time, torch
sys.path.insert(0, '.')
from warp_drive_amd.training.scripts.train import setup_trainer
for graph, dtype, fused in ((False, "float32", True), (True, "float32", True), (True, "float32", False),
                            (True, "bfloat16", False)):
    ov = {"trainer": {"num_envs": 2000, "train_batch_size": 100000, "graph_rollout": graph, "rollout_dtype": dtype,
                      "fused_policy_forward": fused}}
    tr = setup_trainer("tag_continuous", ov, results_dir=f"/tmp/rt{int(graph)}{dtype}{int(fused)}", verbose=False)
    tr._generate_rollout_batch(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): tr._generate_rollout_batch()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    tr.train(1); torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train(2)
    torch.cuda.synchronize()
    it = (time.perf_counter() - t0) / 2
    print(f"graph={graph} rollout_dtype={dtype} fused_policy_forward={fused}: rollout of {tr.batch_len} ticks = {dt*1e3:.1f} ms -> {dt/tr.batch_len*1e3:.3f} ms/tick, "
          f"{tr.train_batch_size/dt:.3e} env-steps/s; training iteration {it*1e3:.0f} ms -> {tr.train_batch_size/it:.3e} env-steps/s end to end")
    tr.graceful_close()
