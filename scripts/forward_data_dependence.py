"""How much of the fused policy forward's time is the DATA (the chip clocks to its power budget): the runner policy of
configs[2] on 200 000 rows with random / zero / 1e-3-scaled weights and random / zero / constant observations.
    python scripts/forward_data_dependence.py        (on the GPU box)"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from warp_drive_amd.managers.function_manager import HIPFunctionManager
from warp_drive_amd.training.models import FullyConnected
from warp_drive_amd.training.policy_kernel import FusedPolicyForward
dev = torch.device("cuda:0")
E, N, F, heads = 2000, 105, 71, [21, 21]
fm = HIPFunctionManager(num_agents=1, num_envs=1); fm.load_hip_from_binary_file()
probs = [torch.zeros(E, N, a, device=dev) for a in heads]
ids = torch.arange(5, 105, dtype=torch.int32, device=dev)
for wmode in ("random", "zero", "small"):
    torch.manual_seed(1)
    model = FullyConnected(F, heads, fc_dims=(256, 256)).to(dev)
    if wmode == "zero":
        for p in model.parameters(): p.data.zero_()
    if wmode == "small":
        for p in model.parameters(): p.data.mul_(1e-3)
    fused = FusedPolicyForward(fm, model, F, arithmetic="bf16x3")
    for omode in ("randn", "zeros", "const"):
        obs = {"randn": torch.randn(E, N, F, device=dev), "zeros": torch.zeros(E, N, F, device=dev), "const": torch.full((E, N, F), 0.5, device=dev)}[omode]
        for _ in range(5): fused(obs, ids, probs)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): fused(obs, ids, probs)
        b.record(); torch.cuda.synchronize()
        print(f"weights {wmode:7s} obs {omode:6s}: {a.elapsed_time(b) / 20 * 1000:.1f} us per launch")
