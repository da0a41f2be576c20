#!/usr/bin/env python3
"""Phase timing of HipTagContinuousTick with s_memtime stamps (WD_TC_PROFILE build).
Run on the GPU box:  python scripts/phase_profile.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "build", "prof")
os.makedirs(out, exist_ok=True)
ablate = os.environ.get("ABLATE", "0")  # WD_TC_ABLATE bits (timing-only variants)
hsaco = os.path.join(out, f"wd_kernels_prof{ablate}.hsaco")
subprocess.run(["hipcc", "--offload-arch=gfx950", "--genco", "-O3", "-std=c++17", "-ffp-contract=off",
                "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-DWD_TC_PROFILE", f"-DWD_TC_ABLATE={ablate}",
                os.path.join(ROOT, "warp_drive_amd/csrc/kernels/wd_kernels.hip"), "-o", hsaco], check=True)
os.environ["WD_HSACO"] = hsaco
os.environ["WD_TC_PROFILE"] = "1"
import numpy as np
import torch

import bench
from warp_drive_amd.env_wrapper import EnvWrapper
from warp_drive_amd.envs.tag_continuous import TagContinuous
from warp_drive_amd.managers.function_manager import HIPSampler
from warp_drive_amd.rollout import RolloutEngine
from warp_drive_amd.training.data_loader import create_and_push_data_placeholders

E = 2000
w = EnvWrapper(env_obj=TagContinuous(**bench.BENCH_CFG), num_envs=E, env_backend="hip")
w.reset_all_envs()
sampler = HIPSampler(w.cuda_function_manager)
sampler.init_random(seed=1)
create_and_push_data_placeholders(env_wrapper=w, action_sampler=sampler, training_batch_size_per_env=None,
                                  push_data_batch_placeholders=False)
for fused in (True, False):
    engine = RolloutEngine(w, sampler, fused=fused)
    import time
    engine.run(50)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    engine.run(2000)
    torch.cuda.synchronize()
    print(f"=== ablate={ablate}: wall per tick {(time.perf_counter() - t0) / 2000 * 1e6:.2f} us")
    raw = w.cuda_data_manager.pull_data_from_device("neighbor_distances").view(np.uint64).reshape(-1, 16)
    n_blocks = engine.plan_grid if hasattr(engine, 'plan_grid') else 2000
    st = raw[:n_blocks].astype(np.int64)
    names = {0: "start", 1: "prologue done", 2: "sampling done", 3: "move (P0) done", 4: "knn A done",
             5: "knn B done", 6: "knn C done", 7: "barrier after knn", 9: "obs gather issued", 10: "rewards done"}
    order = [0, 1, 2, 3, 4, 5, 6, 7, 9, 10]
    print(f"--- {'fused tick' if fused else 'step only'}: mean cycles (s_memtime, 100 MHz ticks?) per phase, wave 0 of each block")
    prev = None
    for k in order:
        if prev is not None:
            d = st[:, k] - st[:, prev]
            print(f"{names[k]:<22} mean={d.mean():10.0f} p10={np.percentile(d,10):9.0f} p90={np.percentile(d,90):9.0f}")
        prev = k
    real = (st[:, 12] - st[:, 11]) * 10.0  # ns (100 MHz constant counter)
    print(f"block lifetime: {real.mean()/1000:.2f} us wall = {(st[:, 10] - st[:, 0]).mean():.0f} shader cycles"
          f" -> shader clock {(st[:, 10] - st[:, 0]).mean() / real.mean():.3f} GHz;"
          f" first start -> last end {(st[:, 12].max() - st[:, 11].min()) / 100.0:.2f} us")
    life = (st[:, 12] - st[:, 11]) / 100.0
    start = (st[:, 11] - st[:, 11].min()) / 100.0
    end = (st[:, 12] - st[:, 11].min()) / 100.0
    pc = lambda a: " ".join(f"{np.percentile(a, q):6.2f}" for q in (0, 10, 50, 90, 99, 100))
    print("percentiles 0/10/50/90/99/100 (us): lifetime", pc(life), "| start", pc(start), "| end", pc(end))
    print("mean lifetime by blockIdx % 8 (XCD):", " ".join(f"{life[i::8].mean():.2f}" for i in range(8)))
    print("mean lifetime by blockIdx quartile:", " ".join(f"{life[q * len(life) // 4:(q + 1) * len(life) // 4].mean():.2f}" for q in range(4)))
    print("mean start by blockIdx quartile:", " ".join(f"{start[q * len(life) // 4:(q + 1) * len(life) // 4].mean():.2f}" for q in range(4)))
    # which phases stretch in the blocks that finish last?  (oldest-first issue arbitration lets the
    # first-launched blocks run ahead; the kernel ends with the slowest)
    ok = (st[:, 4] > st[:, 3]) & (st[:, 7] > st[:, 6])  # wave 0 ran the search (agent 0 in the game)
    order_by_end = np.argsort(st[:, 12])
    fast, slow = order_by_end[: len(order_by_end) // 10], order_by_end[-(len(order_by_end) // 10):]
    for label, idx in (("fastest 10 % of blocks", fast), ("slowest 10 % of blocks", slow)):
        idx = idx[ok[idx]]
        seg = " ".join(f"{names[b][:14]}={np.mean(st[idx, b] - st[idx, a_]) / 1000:5.1f}k"
                       for a_, b in zip(order[:-1], order[1:]))
        print(f"{label}: {seg}  | lifetime {np.mean(st[idx, 12] - st[idx, 11]) / 100:.1f} us")
    tot = st[:, 10] - st[:, 0]
    print(f"{'total':<22} mean={tot.mean():10.0f}   spread of block start = {st[:,0].max()-st[:,0].min()}")

# two replica groups on two HIP streams: do their kernels overlap in time?
engine = RolloutEngine(w, sampler, fused=True, n_groups=2)
engine.run(50)
torch.cuda.synchronize()
raw = w.cuda_data_manager.pull_data_from_device("neighbor_distances").view(np.uint64).reshape(-1, 16)
st = raw[:2000].astype(np.int64)   # 128-thread blocks: one replica per block, group g = rows [1000 g, 1000 (g + 1))
t0 = st[:, 11].min()
for g, sl in enumerate((slice(0, 1000), slice(1000, 2000))):
    print(f"group {g}: last tick blocks start {(st[sl, 11].min() - t0) / 100.0:8.2f} us .. end {(st[sl, 12].max() - t0) / 100.0:8.2f} us "
          f"(mean block lifetime {((st[sl, 12] - st[sl, 11]).mean()) / 100.0:.2f} us)")
