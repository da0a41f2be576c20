#!/usr/bin/env python3
"""Build the oracle's C restatement (test infrastructure): oracle/_build/libwd_oracle.so.

-ffp-contract=off: only the explicit fmaf() calls may fuse (see wd_oracle.c header).
-mfma: make fmaf() a single vfmadd instruction (the hosts numpy dispatches to
AVX2/AVX512+FMA kernels on have it; numpy's float32 cos/sin use FMA there too).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "libwd_oracle.so")
SRC = os.path.join(HERE, "csrc", "wd_oracle.c")


def build(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    if (not force) and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    cmd = ["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-mfma", "-fopenmp",
           "-o", LIB, SRC, "-lm"]
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
