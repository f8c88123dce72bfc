"""Top kernels by total time of a rocprofv3 rocpd database (rocprofv3 --kernel-trace): usage: python tools/rocpd_top.py <results.db> [rows]"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,start,end from kernels order by start").fetchall()
agg = collections.defaultdict(list)
for n, s, e in rows:
    agg[n].append(e - s)
tot = sum(sum(v) for v in agg.values())
print(f"{len(rows)} launches, {tot / 1e6:.1f} ms of kernel time")
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f"{sum(v) / 1e6:9.2f} ms {100 * sum(v) / tot:5.1f} %  {len(v):6d} x {sum(v) / len(v) / 1e3:9.1f} us  {n[:110]}")
