#!/usr/bin/env python3
"""Build and time experimental variants of the kernels WITHOUT switches in the product source.

A variant = the product kernel sources + a list of textual substitutions, compiled into
build/variants/<name>.hsaco and run through bench.py with WD_HSACO pointing at it.

    python experiments/variants.py build  <set>          # here (hipcc cross-compiles)
    python experiments/variants.py bench  <set> [rounds] [bench args...]   # on the GPU box

Variant sets live in experiments/variant_sets.py (name -> [(old, new), ...]); "base" is the product
source unchanged.  Runs are interleaved (boxes of the pool differ by up to 15 % in clock)."""
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "build", "variants")


def build(set_name):
    from experiments.variant_sets import SETS
    from warp_drive_amd import build as wb

    os.makedirs(OUT, exist_ok=True)
    for name, subs in SETS[set_name].items():
        # the patched copy of the sources lives outside the tree (only the code object ships to the GPU box)
        src = os.path.join(tempfile.gettempdir(), "wd_variants", f"src_{name}")
        shutil.rmtree(src, ignore_errors=True)
        shutil.copytree(wb.KDIR, src)
        flags = [new for fname, old, new in subs if fname is None]  # (None, "flag", "-f...") = extra compiler flag
        for fname, old, new in subs:
            if fname is None:
                continue
            path = os.path.join(src, fname)
            text = open(path).read()
            assert text.count(old) >= 1, f"variant {name}: pattern not found in {fname}: {old[:60]!r}"
            open(path, "w").write(text.replace(old, new))
        hsaco = os.path.join(OUT, f"{name}.hsaco")
        cmd = [wb._hipcc(), *wb.KERNEL_FLAGS, *flags, os.path.join(src, "wd_kernels.hip"), "-o", hsaco]
        subprocess.run(cmd, check=True)
        print("built", hsaco)


def bench(set_name, rounds, extra):
    from experiments.variant_sets import SETS

    names = list(SETS[set_name])
    res = {n: [] for n in names}
    for _ in range(rounds):
        for n in names:
            env = dict(os.environ, WD_HSACO=os.path.join(OUT, f"{n}.hsaco"))
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1000", "--warmup", "100",
                                  "--no-cpu-baseline"] + extra, capture_output=True, text=True, env=env)
            try:
                d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
                res[n].append((d["ms_per_step"] * 1e3, d["roofline"]["avg_kernel_us"]))
            except Exception:
                res[n].append((float("nan"), float("nan")))
                print(n, "FAILED", out.stderr[-400:])
    for n in names:
        steps = " ".join(f"{a:7.2f}" for a, _ in res[n])
        kern = " ".join(f"{b:7.2f}" for _, b in res[n])
        print(f"{n:24s} us/step: {steps}   kernel us: {kern}")


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2])
    else:
        bench(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 2, sys.argv[4:])
