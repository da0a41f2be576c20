#!/usr/bin/env python3
"""Exhaustive check (test infrastructure): oracle/csrc/wd_oracle.c:np_sincosf -- the same
routine the device runs (csrc/kernels/wd_common.h:wd_np_sincosf) -- against numpy's float32
cos/sin for EVERY float32 in [0, 2*pi + a few ulps].  ~1.09e9 inputs, a few minutes.
Prints the number of mismatches (expected 0 on x86-64 hosts where numpy dispatches to its
AVX2/AVX512+FMA kernels)."""
import ctypes
import sys

import numpy as np

from oracle import build as obuild

lib = ctypes.CDLL(obuild.build())
hi = int(np.array([2 * np.pi], dtype=np.float32).view(np.int32)[0]) + 16
chunk = 1 << 24
bad_cos = bad_sin = 0
out = np.empty(chunk, dtype=np.float32)
for start in range(0, hi + 1, chunk):
    bits = np.arange(start, min(start + chunk, hi + 1), dtype=np.int32)
    x = bits.view(np.float32)
    o = out[: x.size]
    lib.wdo_np_cosf(x.ctypes.data_as(ctypes.c_void_p), o.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(x.size))
    bad_cos += int((o.view(np.uint32) != np.cos(x).view(np.uint32)).sum())
    lib.wdo_np_sinf(x.ctypes.data_as(ctypes.c_void_p), o.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(x.size))
    bad_sin += int((o.view(np.uint32) != np.sin(x).view(np.uint32)).sum())
print(f"inputs={hi + 1} cos_mismatches={bad_cos} sin_mismatches={bad_sin} numpy={np.__version__}")
sys.exit(1 if (bad_cos or bad_sin) else 0)
