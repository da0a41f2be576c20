#!/usr/bin/env python3
"""Dispatch timeline from a rocprofv3 rocpd database: start / end / gap / queue of the last N
dispatches of kernels matching a substring.  usage: rocpd_timeline.py <db> <substr> [N]"""
import sqlite3
import sys

db, sub = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 24
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info('kernels')")]
qcol = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else None)
scol = "stream_id" if "stream_id" in cols else None
sel = "name, start, end" + (f", {qcol}" if qcol else ", 0") + (f", {scol}" if scol else ", 0")
rows = c.execute(f"select {sel} from kernels where name like ? order by start", (f"%{sub}%",)).fetchall()
rows = rows[-n:]
t0 = rows[0][1]
prev_end = None
print("columns available:", cols)
for name, s, e, q, st in rows:
    gap = "" if prev_end is None else f"{(s - prev_end) / 1e3:8.2f}"
    print(f"{name[:28]:<28} q={q} s={st} start={(s - t0) / 1e3:9.2f} us  dur={(e - s) / 1e3:7.2f} us  gap_after_prev_end={gap}")
    prev_end = e
