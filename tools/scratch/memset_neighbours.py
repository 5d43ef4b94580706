"""Kernel trace (rocpd .db): every hipMemset blit kernel with the launches around it (last third of the trace = steady state)."""
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, start, end from kernels order by start"))
n = len(rows)
lo = n * 2 // 3
short = lambda s: s.split("(")[0][-70:]
hits = [i for i in range(lo, n) if "fillBuffer" in rows[i][0]]
print("kernels %d, memset blits in the last third: %d" % (n, len(hits)))
for i in hits[:8]:
    print("---")
    for j in range(max(lo, i - 3), min(n, i + 3)):
        print("  %s%s  %.1f us" % ("*" if j == i else " ", short(rows[j][0]), (rows[j][2] - rows[j][1]) / 1e3))
