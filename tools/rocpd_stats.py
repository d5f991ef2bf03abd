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
# the GGS kernels are launched in several shapes by one bench run (256-sequence engine passes; the 64-sequence cold-batch leg): per grid
try:
    for r in db.execute(f"select s.kernel_name, d.grid_size_x / d.workgroup_size_x, count(*), avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3, "
                        f"max(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id where s.kernel_name like '%pd_ggs_%kernel%' "
                        f"group by s.kernel_name, d.grid_size_x order by 1, 2").fetchall():
        print(f"# {r[0][:60]:60s} {r[1]:5d} workgroups: {r[2]:5d} launches, average {r[3] / 1e3:8.3f} ms (min {r[4] / 1e3:.3f}, max {r[5] / 1e3:.3f})")
except sqlite3.OperationalError:
    pass
