#!/usr/bin/env python3
"""The fused policy forward alone (runner policy, 200 000 rows of 71 floats, 256-256 trunk, two 21-way heads), both
arithmetics, 10 launches each: the workload of the --pmc / --kernel-trace passes that explain where a tile's cycles go.
    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU ... -- python scripts/policy_forward_pmc.py"""
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
ids = torch.arange(5, 105, dtype=torch.int32, device=dev)
torch.manual_seed(1)
model = FullyConnected(F, heads, fc_dims=(256, 256)).to(dev)
for arithmetic in ("bf16x3", "float32"):
    fused = FusedPolicyForward(fm, model, F, arithmetic=arithmetic)
    for _ in range(10):
        fused(obs, ids, probs)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        fused(obs, ids, probs)
    b.record()
    torch.cuda.synchronize()
    print(f"{arithmetic}: {a.elapsed_time(b) / 10 * 1000:.1f} us per launch (probabilities written, no sampling)")
