"""Co-resident GGS launches from a rocprofv3 kernel trace (rocpd sqlite):  python tools/coresident_from_trace.py results.db [flops_per_launch]

bench.py runs `--pipeline-depth` engine contexts, each on its own HIP stream; their persistent pd_ggs_kernel launches
(one per guided diffusion step; 256 workgroups each by default = one per CU, so launches issued together queue behind each
other workgroup by workgroup; 64 each in the 4 x 64 shape of round 1) overlap in time.  This reads the
kernel dispatch timestamps of a trace of the bench command, merges overlapping pd_ggs_kernel intervals into sets and
reports, per set size, the wall time of a set (union of its intervals) and the fp32-ALU rate it implies
(n launches x flops_per_launch / wall) -- the `roofline.co_resident` figure of bench.py, recomputed from profiles/.
If the tracer serialised the streams every set has size 1 and the tool says so."""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
flops = float(sys.argv[2]) if len(sys.argv) > 2 else 256 * 57000 * 100.0 * 700     # the bench default engine pass
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
def dispatches(pattern):
    q = "select d.start, d.end{g} from {kd} d join {ks} s on d.kernel_id=s.id where s.kernel_name like '%{p}%' order by d.start"
    try:
        return db.execute(q.format(g=", d.grid_size_x", kd=kd, ks=ks, p=pattern)).fetchall()
    except sqlite3.OperationalError:          # a rocpd schema without the grid columns: one shape assumed
        return [(a, b, 0) for a, b in db.execute(q.format(g="", kd=kd, ks=ks, p=pattern)).fetchall()]


# The bench's engine passes run the lane-per-item kernel since round 4 (one workgroup per sequence: 256 workgroups per launch); its
# cold-single-batch leg launches other shapes (64 sequences: the lane kernel on 64 workgroups, the wave-per-item kernel on 4 x 64) with a
# quarter of the FLOPs per launch.  Only the launches of the most time-consuming (kernel, grid) shape are analysed; the rest is listed.
cand = dispatches("pd_ggs_lane_kernel")
which = "pd_ggs_lane_kernel"
others = [("pd_ggs_kernel", r) for r in dispatches("pd_ggs_kernel")]
if not cand:
    cand, others, which = [r for _, r in others], [], "pd_ggs_kernel"
if not cand:
    raise SystemExit("no pd_ggs_kernel / pd_ggs_lane_kernel dispatches in the trace")
by_grid = defaultdict(list)
for r in cand:
    by_grid[r[2]].append(r)
main_grid = max(by_grid, key=lambda g: sum(en - st for st, en, _ in by_grid[g]))
rows = [(st, en) for st, en, _ in by_grid[main_grid]]
others += [(which, r) for g, rs in by_grid.items() if g != main_grid for r in rs]
if main_grid:
    which += f" with a grid of {main_grid} threads"
if others:
    shapes = defaultdict(list)
    for name, (st, en, g) in others:
        shapes[(name, g)].append((en - st) / 1e6)
    for (name, g), od in sorted(shapes.items()):
        print(f"(not counted below: {len(od)} {name} launches with a grid of {g} threads -- another shape: the cold single batch -- average {sum(od) / len(od):.3f} ms)")
sets, cur = [], [rows[0]]
cur_end = rows[0][1]
for st, en in rows[1:]:
    if st < cur_end:                      # overlaps the running set
        cur.append((st, en))
        cur_end = max(cur_end, en)
    else:
        sets.append(cur)
        cur, cur_end = [(st, en)], en
sets.append(cur)
by_size = defaultdict(list)
for s in sets:
    by_size[len(s)].append(s)
durs = [(en - st) / 1e6 for st, en in rows]
print(f"GGS kernel {which}: {len(rows)} launches, average duration {sum(durs) / len(durs):.3f} ms (min {min(durs):.3f}, max {max(durs):.3f})")
print(f"one launch alone: {flops / (sum(durs) / len(durs) * 1e-3) / 1e12:.2f} TFLOP/s = {flops / (sum(durs) / len(durs) * 1e-3) / 1e12 / 157.3 * 100:.1f} % of 157.3 (fp32 vector ALU), "
      f"{flops / 1e9:.2f} GFLOP per launch")
for n in sorted(by_size):
    walls = [(max(e for _, e in s) - min(b for b, _ in s)) / 1e6 for s in by_size[n]]
    w = sum(walls) / len(walls)
    print(f"sets of {n} overlapping launch(es): {len(walls)} sets, wall {w:.3f} ms per set -> {n * flops / (w * 1e-3) / 1e12:.2f} TFLOP/s = "
          f"{n * flops / (w * 1e-3) / 1e12 / 157.3 * 100:.1f} % of the fp32 vector ALU peak")
if max(by_size) == 1 and len(rows) > 4:
    print("NOTE: no two launches overlap in this trace: the tracer serialised the streams; the co-resident figure cannot be read from it "
          "(see the single 256-sequence launch of tools/pmc_target.py 256 1 instead: one launch that fills the chip)")
