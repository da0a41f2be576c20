#!/usr/bin/env python3
"""Summarise rocprofv3 (rocpd sqlite) outputs into the small text/JSON files kept under
profiles/.   usage: rocpd_summary.py kernel <db>      per-kernel calls / avg / min / max (us)
                    rocpd_summary.py pmc <db> <COUNTER>   per-kernel average counter value"""
import json
import sqlite3
import sys


def kernel_stats(db):
    c = sqlite3.connect(db)
    rows = c.execute(
        "select name, count(*), avg(duration), min(duration), max(duration), sum(duration), "
        "max(grid_x), max(workgroup_x), max(lds_size), max(vgpr_count), max(sgpr_count) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[5] for r in rows) or 1
    out = []
    for r in rows:
        out.append({"kernel": r[0], "calls": r[1], "avg_us": r[2] / 1e3, "min_us": r[3] / 1e3,
                    "max_us": r[4] / 1e3, "pct": 100.0 * r[5] / total, "grid_x": r[6], "wg_x": r[7],
                    "lds": r[8], "vgpr": r[9], "sgpr": r[10]})
    return out


def pmc_stats(db, counter):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info('counters_collection')")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    q = (f"select {name_col}, counter_name, count(*), avg(value), min(value), max(value) from counters_collection "
         f"where counter_name = ? group by {name_col}, counter_name order by avg(value) desc")
    return [{"kernel": r[0], "counter": r[1], "dispatches": r[2], "avg": r[3], "min": r[4], "max": r[5]}
            for r in c.execute(q, (counter,))]


if __name__ == "__main__":
    if sys.argv[1] == "kernel":
        rows = kernel_stats(sys.argv[2])
        print(f"{'kernel':<40}{'calls':>8}{'avg_us':>10}{'min_us':>10}{'max_us':>10}{'pct':>7}{'grid':>7}{'wg':>5}{'lds':>7}{'vgpr':>5}")
        for r in rows:
            print(f"{r['kernel'][:39]:<40}{r['calls']:>8}{r['avg_us']:>10.2f}{r['min_us']:>10.2f}{r['max_us']:>10.2f}"
                  f"{r['pct']:>7.1f}{r['grid_x']:>7}{r['wg_x']:>5}{r['lds']:>7}{r['vgpr']:>5}")
    else:
        print(json.dumps(pmc_stats(sys.argv[2], sys.argv[3]), indent=1))
