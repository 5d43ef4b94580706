"""Summarise a rocprofv3 rocpd database (kernel trace) into a small per-kernel CSV restricted to the last `steps`
train steps (steps are delimited by the fused-Adam launches), so MIOpen's first-call search kernels are excluded.
usage: prof_summary.py <dir-or-db> <out.csv> [steps]"""
import csv, glob, sqlite3, sys
src, dst = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 0
f = glob.glob(src + "/*.db")[0] if not src.endswith(".db") else src
cur = sqlite3.connect(f).cursor()
t0, t1 = cur.execute("select min(start), max(end) from kernels").fetchone()
if steps:
    adam = [r[0] for r in cur.execute("select end from kernels where name like '%adam%' or name like '%Adam%' or name like '%FusedOptimizer%' order by end")]
    groups = []
    for e in adam:
        if not groups or e - groups[-1] > 5e6:      # > 5 ms apart -> next step
            groups.append(e)
        else:
            groups[-1] = e
    if len(groups) > steps:
        t0, t1 = groups[-steps - 1], groups[-1]
rows = cur.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                   "from kernels where start >= ? and end <= ? group by name order by 3 desc", (t0, t1)).fetchall()
tot = sum(r[2] for r in rows)
n = max(steps, 1)
with open(dst, "w", newline="") as out:
    w = csv.writer(out)
    w.writerow(["kernel", "calls_per_step", "ms_per_step", "avg_us", "min_us", "max_us", "pct"])
    for r in rows:
        w.writerow([r[0][:120].replace(",", ";"), "%.1f" % (r[1] / n), "%.3f" % (r[2] / n), "%.2f" % r[3], "%.2f" % r[4], "%.2f" % r[5],
                    "%.2f" % (100 * r[2] / tot)])
print("window %.1f ms, kernel time %.2f ms per step over %d kernels" % ((t1 - t0) / 1e6 / n, tot / n, len(rows)))
