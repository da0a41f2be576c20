import sys, time, torch
sys.path.insert(0, '.')
from warp_drive_amd.training.scripts.train import setup_trainer
for graph in (True, False):
    ov = {"trainer": {"num_envs": 2000, "train_batch_size": 100000, "graph_rollout": graph}}
    tr = setup_trainer("tag_continuous", ov, results_dir=f"/tmp/rt{int(graph)}", verbose=False)
    tr._generate_rollout_batch(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): tr._generate_rollout_batch()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"graph={graph}: rollout of {tr.batch_len} ticks = {dt*1e3:.1f} ms -> {dt/tr.batch_len*1e3:.3f} ms/tick, {tr.train_batch_size/dt:.3e} env-steps/s")
    tr.graceful_close()
