#!/usr/bin/env python3
"""Summarises a rocprofv3 rocpd SQLite database (the default output of `rocprofv3 --kernel-trace --stats`) into a
per-kernel table: calls, total / average / min / max duration.  Usage: rocpd_stats.py results.db [out.md]"""
import sqlite3, sys

def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute(f"pragma table_info({disp})")]
    scols = [r[1] for r in cur.execute(f"pragma table_info({sym})")]
    name_col = "kernel_name" if "kernel_name" in scols else "display_name"
    rows = cur.execute(f"select s.{name_col}, d.start, d.end from {disp} d join {sym} s on d.kernel_id = s.id").fetchall()
    agg = {}
    for name, s, e in rows:
        short = name.split("(")[0].replace("void ", "").replace("jxlhip::", "")
        a = agg.setdefault(short, [0, 0, 1 << 62, 0])
        dt = e - s
        a[0] += 1; a[1] += dt; a[2] = min(a[2], dt); a[3] = max(a[3], dt)
    total = sum(a[1] for a in agg.values()) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"| {k} | {a[0]} | {a[1] / 1e6:.3f} | {a[1] / a[0] / 1e3:.1f} | {a[2] / 1e3:.1f} | {a[3] / 1e3:.1f} | {100 * a[1] / total:.1f} |")
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "a").write(out)
    print(out)

if __name__ == "__main__":
    main()
