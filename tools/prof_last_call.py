"""Print the kernels of the last forward+backward call in a rocprofv3 rocpd db, in launch order: prof_last_call.py <dir> [n]"""
import glob, sqlite3, sys
f = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
cur = sqlite3.connect(f).cursor()
rows = cur.execute("select name, start, end from kernels order by start desc limit ?", (n,)).fetchall()[::-1]
t0 = rows[0][1]
prev = None
for name, s, e in rows:
    gap = 0 if prev is None else (s - prev) / 1e3
    print("%-60s start %9.1f us  dur %8.1f us  gap %6.1f" % (name.replace("(anonymous namespace)::", "").replace("void ", "")[:60], (s - t0) / 1e3, (e - s) / 1e3, gap))
    prev = e
print("span %.1f us" % ((rows[-1][2] - t0) / 1e3))
