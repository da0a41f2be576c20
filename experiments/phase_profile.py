#!/usr/bin/env python3
"""Per-wavefront phase timeline of the fused TagContinuous tick (variant "prof" of
experiments/variant_sets.py: s_memtime stamps at the phase boundaries, written through a
__device__ pointer the harness sets).  Run on the GPU box after `variants.py build profile`:
    python experiments/phase_profile.py [variant-name] [num_envs] [episode tick of the stamped launch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
name = sys.argv[1] if len(sys.argv) > 1 else "prof"
E = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
T_STAMP = int(sys.argv[3]) if len(sys.argv) > 3 else 300  # (episodes are 500 ticks; the live-agent count falls along them)
os.environ["WD_HSACO"] = os.path.join(ROOT, "build", "variants", f"{name}.hsaco")
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import time

import numpy as np
import torch

import bench
from warp_drive_amd.env_wrapper import EnvWrapper
from warp_drive_amd.envs.tag_continuous import TagContinuous
from warp_drive_amd.managers import hip_driver as drv
from warp_drive_amd.managers.function_manager import HIPSampler
from warp_drive_amd.rollout import RolloutEngine
from warp_drive_amd.training.data_loader import create_and_push_data_placeholders

w = EnvWrapper(env_obj=TagContinuous(**bench.BENCH_CFG), num_envs=E, env_backend="hip")
w.reset_all_envs()
sampler = HIPSampler(w.cuda_function_manager)
sampler.init_random(seed=1)
create_and_push_data_placeholders(env_wrapper=w, action_sampler=sampler, training_batch_size_per_env=None,
                                  push_data_batch_placeholders=False)
engine = RolloutEngine(w, sampler, fused=True)
n_waves = 2 * E
buf = drv.mem_alloc(n_waves * 16 * 8)
drv.memset(buf, 0, n_waves * 16 * 8)
sym, nbytes = w.cuda_function_manager._module.get_global("tc_prof_g")
drv.memcpy_htod(sym, np.array([int(buf)], dtype=np.uint64))
engine.run(300)
torch.cuda.synchronize()
t0 = time.perf_counter()
engine.run(1000)
torch.cuda.synchronize()
print(f"=== {name}: wall per tick {(time.perf_counter() - t0) / 1000 * 1e6:.2f} us (stamped build)")
engine.run((T_STAMP - 1300) % 500)
torch.cuda.synchronize()
live = w.cuda_data_manager.pull_data_from_device("still_in_the_game").sum(axis=1).mean()
print(f"stamped launch = tick {T_STAMP} of an episode, {live:.1f} agents in the game")
drv.memset(buf, 0, n_waves * 16 * 8)
torch.cuda.synchronize()
engine.run(1)
torch.cuda.synchronize()
raw = np.zeros(n_waves * 16, dtype=np.uint64)
drv.memcpy_dtoh(raw, buf)
drv.synchronize()
st = raw.reshape(-1, 16).astype(np.int64)
fell_back = st[:, 8] > 0
print(f"wavefronts with a lane outside the in-order exit of the search (exact ranking branch): {fell_back.sum()} of {(st[:, 13] > 0).sum()}")
searched = st[:, 7] > 0
print(f"wavefronts that ran the search: {searched.sum()} of {(st[:, 13] > 0).sum()}")
st[:, 7] = np.where(searched, st[:, 7], st[:, 6])
st[:, 8] = np.where(fell_back, st[:, 8], st[:, 7])
names = ["start", "loads issued+tables", "sampled", "barrier1", "moved", "barrier2", "tags", "knn key chain", "(ranking branch entered)", "knn ids / ranking branch",
         "ids flushed", "obs gathered+flushed", "barrier3", "rewards/end"]
for wv, label in ((0, "wave 0 of each block (64 agents)"), (1, "wave 1 of each block (41 agents)")):
    s = st[wv::2]
    ok = (s[:, 6] > 0) & (s[:, 13] > 0)
    s = s[ok]
    print(f"--- {label}: {ok.sum()} waves; mean / p10 / p90 shader cycles per phase")
    for k in range(1, 14):
        d = s[:, k] - s[:, k - 1]
        print(f"  {names[k]:<24} {d.mean():9.0f} {np.percentile(d, 10):9.0f} {np.percentile(d, 90):9.0f}")
    tot = s[:, 13] - s[:, 0]
    real = (s[:, 15] - s[:, 14]) * 10.0
    print(f"  total {tot.mean():.0f} cycles = {real.mean() / 1000:.2f} us -> {tot.mean() / real.mean():.3f} GHz")
s = st[(st[:, 14] > 0) & (st[:, 15] > 0)]
t0 = s[:, 14].min()
start, end = (s[:, 14] - t0) / 100.0, (s[:, 15] - t0) / 100.0
pc = lambda a: " ".join(f"{np.percentile(a, q):6.2f}" for q in (0, 10, 50, 90, 99, 100))
print("percentiles 0/10/50/90/99/100 (us): wave start", pc(start), "| wave end", pc(end), "| lifetime", pc(end - start))
# absolute timeline of the phase boundaries (us since the first wave started), wave 0 only
s0 = st[0::2]
s0 = s0[(s0[:, 6] > 0) & (s0[:, 13] > 0)]
ghz = ((s0[:, 13] - s0[:, 0]) / ((s0[:, 15] - s0[:, 14]) * 10.0)).mean()
base = (s0[:, 14] - t0) / 100.0
print("phase boundary, us since first wave start (p10 / p50 / p90), wave 0:")
for k in range(0, 14):
    tk = base + (s0[:, k] - s0[:, 0]) / (ghz * 1000.0)
    print(f"  {names[k]:<24} {np.percentile(tk, 10):7.2f} {np.percentile(tk, 50):7.2f} {np.percentile(tk, 90):7.2f}")
