#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel stats and the
per-launch timeline of the last frame.  usage: rocpd_summary.py results.db [--launches]"""
import re
import sqlite3
import sys


def short(name):
    m = re.search(r"(k_[a-z_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, start, end, grid_x, workgroup_x, vgpr_count, lds_size from kernels order by start").fetchall()
    agg = {}
    for name, st, en, gx, wx, vg, lds in rows:
        k = short(name)
        a = agg.setdefault(k, [0, 0.0, 1e30, 0.0, vg, lds])
        d = (en - st) / 1e3
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    print(f"{'kernel':34s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s} vgpr lds")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:34s} {a[0]:6d} {a[1]:12.1f} {a[1]/a[0]:10.1f} {a[2]:10.1f} {a[3]:10.1f} {100*a[1]/total:6.1f} {a[4]} {a[5]}")
    print(f"{'TOTAL':34s} {len(rows):6d} {total:12.1f}")
    if "--launches" in sys.argv:
        n = int(sys.argv[sys.argv.index("--launches") + 1]) if len(sys.argv) > sys.argv.index("--launches") + 1 else 60
        t0 = rows[-n][1] if len(rows) >= n else rows[0][1]
        prev_end = None
        for name, st, en, gx, wx, vg, lds in rows[-n:]:
            gap = (st - prev_end) / 1e3 if prev_end else 0.0
            print(f"  +{(st - t0)/1e3:10.1f} us  dur {(en - st)/1e3:9.1f} us  gap {gap:7.1f}  grid {gx//max(1,wx):6d} blk  {short(name)}")
            prev_end = en


if __name__ == "__main__":
    main()
