"""Per-kernel totals from a rocprofv3 (rocpd) sqlite database: python tools/rocpd_stats.py results.db [top]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 16
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute(f"select s.kernel_name, count(*), sum(d.end-d.start)/1e3, avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3, max(d.end-d.start)/1e3 "
                  f"from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"{'kernel':80s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>8s} {'min_us':>8s} {'max_us':>8s} {'%':>6s}")
for r in rows[:top]:
    print(f"{r[0][:80]:80s} {r[1]:6d} {r[2] / 1e3:9.3f} {r[3]:8.1f} {r[4]:8.1f} {r[5]:8.1f} {100 * r[2] / tot:6.1f}")
