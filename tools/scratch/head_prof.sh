cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_head
N=6 rocprofv3 --kernel-trace -d /tmp/prof_head -- python3 $R/tools/scratch/head_only.py > /dev/null 2>&1
python3 - <<PY
import glob, sqlite3, collections
f = glob.glob("/tmp/prof_head/**/*.db", recursive=True)[0]
cur = sqlite3.connect(f).cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
# iterations: split at the focal... use the last 3 of 6 iterations by time: find head_rows_fwd_kernel starts
starts = [s for n, s, e in rows if "head_rows_fwd_kernel" in n]
t0 = starts[-3] - 3_000_000   # ~3 ms before: start of that forward (approximate)
sel = [(n, e - s) for n, s, e in rows if s >= t0]
agg = collections.defaultdict(lambda: [0, 0])
for n, d in sel:
    agg[n][0] += 1; agg[n][1] += d
tot = sum(v[1] for v in agg.values())
print("last 3 iterations: %d kernels, %.3f ms kernel time per iteration" % (len(sel) // 3, tot / 3e6))
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%6.1f %8.1f us  %s" % (c / 3, d / 3e3, n.replace("(anonymous namespace)::", "").replace("void ", "").replace("at::native::", "")[:110]))
PY
