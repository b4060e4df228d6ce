#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel trace): per (kernel, grid) count / avg / total.
usage: rocpd_summary.py run_results.db [--by-grid] [--skip N first dispatches]"""
import re
import sqlite3
import sys


def main():
    path = sys.argv[1]
    by_grid = "--by-grid" in sys.argv
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")]
    if not disp or not sym:     # a profiled command that died before its first dispatch leaves a database without these tables
        print(f"{path}: no kernel-dispatch table ({len(tabs)} tables) -- the profiled command launched no kernel; see its cmd.log")
        return 0
    disp, sym = disp[0], sym[0]
    cols = [r[1] for r in cur.execute(f"pragma table_info({disp})")]
    scols = [r[1] for r in cur.execute(f"pragma table_info({sym})")]
    namecol = "kernel_name" if "kernel_name" in scols else "display_name"
    gx = "grid_size_x" if "grid_size_x" in cols else "grid_x"
    q = f"select s.{namecol}, d.{gx}, d.grid_size_y, d.grid_size_z, d.workgroup_size_x, d.start, d.end from {disp} d join {sym} s on d.kernel_id = s.id order by d.start"
    rows = list(cur.execute(q))
    agg = {}
    for name, g0, g1, g2, wg, st, en in rows:
        short = re.sub(r"\(anonymous namespace\)::", "", name)
        short = re.sub(r"\(.*", "", short)
        key = (short, (g0 // max(wg, 1), g1, g2)) if by_grid else (short,)
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += (en - st) / 1e3
    if not rows:
        print(f"{path}: kernel-dispatch table is empty")
        return 0
    tot = sum(a[1] for a in agg.values())
    print(f"{'kernel':90s} {'n':>6s} {'avg_us':>9s} {'total_us':>11s} {'%':>6s}")
    for key, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        label = key[0][:70] + (f" grid{key[1]}" if by_grid else "")
        print(f"{label:90s} {n:6d} {t / n:9.2f} {t:11.1f} {100 * t / tot:6.2f}")
    print(f"total kernel time {tot / 1e3:.3f} ms over {len(rows)} dispatches")


if __name__ == "__main__":
    main()
