#!/usr/bin/env python3
"""Where a block of the fused policy forward spends its time (variant "mlp_prof" of
experiments/variant_sets.py: s_memtime stamps per wavefront).  Run on the GPU box after
`python experiments/variants.py build mlp_profile`."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["WD_HSACO"] = os.path.join(ROOT, "build", "variants", "mlp_prof.hsaco")
import numpy as np
import torch

from warp_drive_amd.managers import hip_driver as drv
from warp_drive_amd.managers.function_manager import HIPFunctionManager
from warp_drive_amd.training.models import FullyConnected
from warp_drive_amd.training.policy_kernel import FusedPolicyForward

dev = torch.device("cuda:0")
fm = HIPFunctionManager(num_agents=1, num_envs=1)
fm.load_hip_from_binary_file()
E, N, F, heads = 2000, 105, 71, [21, 21]
model = FullyConnected(F, heads, fc_dims=(256, 256)).to(dev)
fused = FusedPolicyForward(fm, model, F)
obs = torch.randn(E, N, F, device=dev)
probs = [torch.zeros(E, N, a, device=dev) for a in heads]
ids = torch.arange(5, 105, dtype=torch.int32, device=dev)
n_waves = (E * 100 + 31) // 32
buf = drv.mem_alloc(n_waves * 8 * 8)
drv.memset(buf, 0, n_waves * 8 * 8)
sym, _ = fm._module.get_global("mlp_prof_g")
drv.memcpy_htod(sym, np.array([int(buf)], dtype=np.uint64))
for _ in range(3):
    fused(obs, ids, probs)
torch.cuda.synchronize()
raw = np.zeros(n_waves * 8, dtype=np.uint64)
drv.memcpy_dtoh(raw, buf)
drv.synchronize()
st = raw.reshape(-1, 8).astype(np.int64)
st = st[(st[:, 6] > 0)]
names = ["row loads issued", "first chunk + rows landed", "layer 1 (384 MFMAs)", "layer 2 (1024 MFMAs)", "output layer (256 MFMAs)",
         "softmax", "stores"]
print(f"{len(st)} wavefronts; mean / p10 / p90 shader cycles")
for k in range(1, 7):
    d = st[:, k] - st[:, k - 1]
    print(f"  {names[k]:<28} {d.mean():9.0f} {np.percentile(d, 10):9.0f} {np.percentile(d, 90):9.0f}")
tot = st[:, 6] - st[:, 0]
print(f"  total {tot.mean():.0f} cycles; the MFMAs alone: {1664 * 64} cycles")
