#!/usr/bin/env python3
"""What does the HBM write path give to "B blocks, each streaming its own contiguous slice"?
(the rollout's store pattern: one replica's observation rows per block) vs a plain fill."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from warp_drive_amd.managers import hip_driver as drv  # noqa: E402
from warp_drive_amd.managers.function_manager import HIPFunctionManager  # noqa: E402

drv.init(0)
fm = HIPFunctionManager(num_agents=1, num_envs=1)
fm.load_hip_from_binary_file()
fm.initialize_functions(["wd_write_probe"])
probe = fm.get_function("wd_write_probe")


def run(blocks, floats_per_block, threads, vec, reps=200):
    stride = (floats_per_block + 3) // 4 * 4
    out = torch.empty(blocks * stride, dtype=torch.float32, device="cuda")
    args = (out, np.int64(stride), np.int32(floats_per_block), np.int32(vec), np.float32(1.0))
    for _ in range(10):
        probe(*args, block=(threads, 1, 1), grid=(blocks, 1))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        probe(*args, block=(threads, 1, 1), grid=(blocks, 1))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return dt, 4.0 * blocks * floats_per_block / dt / 1e12


print(f"{'blocks':>7} {'KB/block':>9} {'threads':>8} {'vec':>4} {'us':>8} {'TB/s':>6}")
for blocks, fpb in ((2000, 7455), (2000, 76545), (16000, 7455), (2000, 7455 * 8), (256, 7455 * 8), (256, 76545 * 8),
                    (65536, 4096)):
    for threads in (128, 256):
        for vec in (1, 4):
            dt, tbs = run(blocks, fpb, threads, vec)
            print(f"{blocks:>7} {fpb * 4 / 1024:>9.1f} {threads:>8} {vec:>4} {dt * 1e6:>8.1f} {tbs:>6.2f}")
