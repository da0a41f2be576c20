#!/usr/bin/env python3
"""Rollout policy forward at the BASELINE shape (2000 replicas x 100 runners / 5 taggers, 71-float rows,
256-256 trunk, two 21-way heads): the fused kernel (csrc/kernels/policy_mlp.hip) against the
framework path the trainer used before (index_select + forward_inference + index_copy_ into the
sampler's tensors + the batch copy of the rows).  Run on the GPU box."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from warp_drive_amd.managers.function_manager import HIPFunctionManager
from warp_drive_amd.training.models import FullyConnected
from warp_drive_amd.training.policy_kernel import FusedPolicyForward

dev = torch.device("cuda:0")
E, N, F, heads = 2000, 105, 71, [21, 21]
fm = HIPFunctionManager(num_agents=1, num_envs=1)
fm.load_hip_from_binary_file()
obs = torch.randn(E, N, F, device=dev)
probs = [torch.zeros(E, N, a, device=dev) for a in heads]
T = 4
row = torch.tensor(1, dtype=torch.int64, device=dev)


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1000.0


for label, ids in (("runners (200 000 rows)", list(range(5, 105))), ("taggers (10 000 rows)", list(range(5)))):
    torch.manual_seed(1)
    model = FullyConnected(F, heads, fc_dims=(256, 256)).to(dev)
    ids_t = torch.tensor(ids, dtype=torch.int32, device=dev)
    ids_l = ids_t.long()
    batch_obs = torch.zeros(T, E, len(ids), F, device=dev)
    fused = FusedPolicyForward(fm, model, F)

    def framework(dtype=None):
        obs_p = obs.index_select(1, ids_l)
        batch_obs.index_copy_(0, row.reshape(1), obs_p.unsqueeze(0))
        ps, _ = model.forward_inference(obs_p, dtype=dtype)
        for h, p in enumerate(ps):
            probs[h].index_copy_(1, ids_l, p)

    t_f32 = timed(framework)
    t_bf16 = timed(lambda: framework(torch.bfloat16))
    per_waves = {}
    for wpb in (1, 2, 4):
        fused.WAVES_PER_BLOCK = wpb
        per_waves[wpb] = timed(lambda: fused(obs, ids_t, probs, obs_out=batch_obs, batch_row=row))
    fused.WAVES_PER_BLOCK = FusedPolicyForward.WAVES_PER_BLOCK
    t_fused = timed(lambda: fused(obs, ids_t, probs, obs_out=batch_obs, batch_row=row))
    t_pack = timed(fused.pack, n=10)
    rows = E * len(ids)
    flops = 2.0 * rows * (F * 256 + 256 * 256 + 256 * 43)
    print(f"{label}: framework fp32 {t_f32:8.1f} us | framework bf16 GEMMs {t_bf16:8.1f} us | fused fp32 MFMA "
          f"{t_fused:8.1f} us ({flops / t_fused / 1e6:.1f} TFLOP/s useful; wavefronts per block 1/2/4: "
          f"{per_waves[1]:.0f}/{per_waves[2]:.0f}/{per_waves[4]:.0f} us) | re-pack after a weight update {t_pack:.0f} us")
