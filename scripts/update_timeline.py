#!/usr/bin/env python3
"""Kernels of ONE update in launch order (name, duration; gaps in which no kernel ran) from a rocprofv3 --kernel-trace database of
scripts/update_profile.py: python scripts/update_timeline.py <db> [min_us]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
rows = c.execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
# the last update = everything after the last HipPolicyGradientHead pair's predecessor: take the tail after the 2nd-to-last objective of the first policy
idx = [i for i, r in enumerate(rows) if r[0].startswith("HipPolicyGradientHead")]
start = idx[-2] if len(idx) >= 2 else 0
t0 = rows[start][1]
prev_end = None
for name, s, e, gx, wg in rows[start:]:
    d = (e - s) / 1e3
    if prev_end is not None and (s - prev_end) / 1e3 >= min_us:
        print(f"{(prev_end - t0) / 1e6:9.3f} ms  {(s - prev_end) / 1e3:9.1f} us  -- no kernel running --")
    prev_end = max(prev_end or e, e)
    if d >= min_us:
        print(f"{(s - t0) / 1e6:9.3f} ms  {d:9.1f} us  grid {gx:>9} wg {wg:>4}  {name[:70]}")
