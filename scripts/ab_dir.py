#!/usr/bin/env python3
"""A/B of code-object variants on the GPU box:  python scripts/ab_dir.py <rounds> <name> [<name> ...] [-- bench.py args]

A variant = a directory build/variants/<name>/ holding replacement code objects (any subset of the files of
warp_drive_amd/csrc/*.hsaco, same file names); bench.py runs with WD_HSACO_DIR pointing at it, so everything the
variant does not replace is the product's.  Runs are interleaved (boxes of the pool differ by a few per cent in
clock).  One line per variant: us per step of every run, then the HIP-event kernel average of every run."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if "--" in args:
    k = args.index("--")
    args, extra = args[:k], args[k + 1:]
rounds, names = int(args[0]), args[1:]
res = {n: [] for n in names}
for _ in range(rounds):
    for n in names:
        vdir = os.path.join(ROOT, "build", "variants", n)
        env = dict(os.environ, WD_HSACO_DIR=vdir)
        if os.path.exists(os.path.join(vdir, "env")):  # KEY=VALUE lines: host-side switches that belong to the variant
            env.update(dict(line.strip().split("=", 1) for line in open(os.path.join(vdir, "env")) if "=" in line))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1000", "--warmup", "100",
                              "--no-cpu-baseline", "--no-spread"] + extra, capture_output=True, text=True, env=env)
        try:
            d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
            res[n].append((d["ms_per_step"] * 1e3, d["roofline"]["avg_kernel_us"], d["roofline"].get("full_load_us") or 0.0))
        except Exception:
            res[n].append((float("nan"),) * 3)
            sys.stderr.write(out.stderr[-2000:])
for n in names:
    print(f"{n:<24} us/step: " + " ".join(f"{a:7.2f}" for a, _, _ in res[n]) + "   kernel us: " +
          " ".join(f"{b:7.2f}" for _, b, _ in res[n]) + "   full-load us: " + " ".join(f"{c:7.2f}" for _, _, c in res[n]))
