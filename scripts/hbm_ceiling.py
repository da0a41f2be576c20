#!/usr/bin/env python3
"""Practical HBM ceilings on this box with plain torch kernels: fill (write only), sum (read only),
copy (read + write).  Context for the roofline fractions in DESIGN.md (peak = 8 TB/s spec)."""
import time

import torch

n = 1 << 28  # 1 GiB of float32
x = torch.empty(n, dtype=torch.float32, device="cuda")
y = torch.empty_like(x)
for name, fn, nbytes in (("fill  (write)", lambda: x.fill_(1.0), 4 * n), ("sum   (read)", lambda: x.sum(), 4 * n),
                         ("copy  (read+write)", lambda: y.copy_(x), 8 * n)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"{name:<20} {nbytes / dt / 1e12:6.2f} TB/s")
