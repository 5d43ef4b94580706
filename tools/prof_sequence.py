"""Kernel SEQUENCE of the last train step of a rocprofv3 kernel trace, run-length compressed by a short kernel label:
usage: prof_sequence.py <dir-or-db> <out.txt>.  One line per run: start offset (ms), count, total us, label."""
import glob, re, sqlite3, sys
src, dst = sys.argv[1], sys.argv[2]
f = glob.glob(src + "/*.db")[0] if not src.endswith(".db") else src
cur = sqlite3.connect(f).cursor()
adam = [r[0] for r in cur.execute("select end from kernels where name like '%adam%' or name like '%Adam%' order by end")]
groups = []
for e in adam:
    if not groups or e - groups[-1] > 5e6:
        groups.append(e)
    else:
        groups[-1] = e
t0, t1 = groups[-2], groups[-1]


def label(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").replace("at::native::", "")
    m = re.search(r"(\w+Functor\w*|\w+_kernel_cuda|\w+kernel_impl\w*)", n)
    head = n.split("(")[0].split("<")[0]
    if head in ("vectorized_elementwise_kernel", "elementwise_kernel_manual_unroll", "elementwise_kernel", "unrolled_elementwise_kernel",
                "reduce_kernel", "index_elementwise_kernel") and m:
        inner = re.findall(r"(\w+)(?:Functor|_kernel_cuda|_kernel_impl)", n)
        return head[:12] + ":" + (inner[0] if inner else m.group(1))[:40]
    return head[:60]


rows = cur.execute("select name, start, end from kernels where start >= ? and end <= ? order by start", (t0, t1)).fetchall()
out, runs = open(dst, "w"), []
for name, s, e in rows:
    l = label(name)
    if runs and runs[-1][0] == l:
        runs[-1][2] += 1
        runs[-1][3] += (e - s) / 1e3
    else:
        runs.append([l, (s - t0) / 1e6, 1, (e - s) / 1e3])
for l, off, c, us in runs:
    out.write("%8.3f  x%-3d %9.1f us  %s\n" % (off, c, us, l))
print(len(rows), "kernels,", len(runs), "runs")
