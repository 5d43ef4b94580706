"""Idle gaps between consecutive kernels in the last train step of a rocprofv3 rocpd db: prof_gaps.py <dir> [min_gap_us]
Steps are delimited by the fused-AdamW launches, the same rule as prof_summary.py (round 3 delimited them by every
`multi_tensor_apply` launch; the BN running-statistics updates and the gradient-norm reductions are multi-tensor launches in
the MIDDLE of a step, so the window was cut short: 38.2 ms / 1 041 kernels next to a summary of 43.1 ms / 1 460)."""
import glob, sqlite3, sys
f = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
cur = sqlite3.connect(f).cursor()
adam = [r[0] for r in cur.execute("select end from kernels where name like '%adam%' or name like '%Adam%' or name like '%FusedOptimizer%' order by end")]
groups = []
for e in adam:
    if not groups or e - groups[-1] > 5e6:
        groups.append(e)
    else:
        groups[-1] = e
t0, t1 = groups[-2], groups[-1]
rows = cur.execute("select name, start, end from kernels where start >= ? and end <= ? order by start", (t0, t1)).fetchall()
busy = sum(e - s for _, s, e in rows)
print("step window %.2f ms, kernel time %.2f ms, %d kernels" % ((t1 - t0) / 1e6, busy / 1e6, len(rows)))
gaps = []
prev_end, prev_name = t0, "(previous step)"
for name, s, e in rows:
    if s - prev_end > 0:
        gaps.append((s - prev_end, prev_name, name, (s - t0) / 1e6))
    if e > prev_end:
        prev_end, prev_name = e, name
tot = sum(g[0] for g in gaps)
print("total idle %.2f ms in %d gaps; gaps >= %.0f us:" % (tot / 1e6, len(gaps), thr))
small = sum(g[0] for g in gaps if g[0] < thr * 1e3)
print("  sum of gaps below threshold: %.2f ms" % (small / 1e6))
for g in sorted(gaps, reverse=True)[:25]:
    if g[0] >= thr * 1e3:
        print("  %8.1f us at %7.2f ms  after %-45s before %s" % (g[0] / 1e3, g[3], g[1].replace("(anonymous namespace)::", "")[:45], g[2].replace("(anonymous namespace)::", "")[:45]))
# idle time and kernel count per millisecond of the step: where the gaps sit
nb = int((t1 - t0) / 1e6) + 1
idle, cnt = [0.0] * nb, [0] * nb
for gdur, _, _, at in gaps:
    idle[min(int(at), nb - 1)] += gdur / 1e3
for _, s, _ in rows:
    cnt[min(int((s - t0) / 1e6), nb - 1)] += 1
print("per ms of the step: idle us / kernels started")
print("  " + " ".join("%d:%d/%d" % (i, idle[i], cnt[i]) for i in range(nb)))
# memory operations inside the window, if the trace has them
try:
    mem = cur.execute("select name, start, end from memory_copies where start >= ? and end <= ? order by start", (t0, t1)).fetchall()
    print("memory copies in the window: %d, %.1f us" % (len(mem), sum(e - s for _, s, e in mem) / 1e3))
except Exception as exc:
    print("no memory-copy table:", exc)
