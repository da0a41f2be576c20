#!/usr/bin/env python3
"""Ten blocks per CU for the fused TagContinuous tick (variant set "occupancy": 11 staging rows per wavefront, LDS
15.9 KB per block): bench.py with the host's staging target patched to match the variant's.
    python experiments/variants.py build occupancy            # here
    python experiments/occupancy_variant.py lds16k --num-envs 16000 --steps 1000 --warmup 100 --no-cpu-baseline --no-spread   # GPU box
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
name = sys.argv[1]
os.environ["WD_HSACO"] = os.path.join(ROOT, "build", "variants", f"{name}.hsaco")
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
import bench  # noqa: E402
from warp_drive_amd.envs.tag_continuous import TagContinuous  # noqa: E402

if name != "base":
    TagContinuous.STAGE_TARGET_BYTES = 3300
bench.main()
