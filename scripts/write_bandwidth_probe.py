#!/usr/bin/env python3
"""What a store-only launch can write on this GPU: torch fill_ / copy_ of large float32 tensors (HBM-resident: 1 - 4 GB) and the
same at the size of one Cartpole 50-tick record (140 MB: inside the Infinity Cache).  Context for configs[4]: the fused
Cartpole rollout reads almost nothing and streams 28-byte records (profiles/r06_bench_cartpole_T50.json)."""
import torch

dev = torch.device("cuda:0")


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3


for mb in (140, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, dtype=torch.float32, device=dev)
    y = torch.empty(n, dtype=torch.float32, device=dev)
    t = timed(lambda: x.fill_(1.0))
    print(f"fill_ {mb:5d} MiB: {mb * 2**20 / t / 1e12:7.2f} TB/s written")
    t = timed(lambda: y.copy_(x))
    print(f"copy_ {mb:5d} MiB: {mb * 2**20 / t / 1e12:7.2f} TB/s written + the same read")
    del x, y
