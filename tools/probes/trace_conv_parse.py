import sys, csv, glob, statistics
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
groups, cur = [], []
for name, us in seq:
    if "conv3x3_direct_bf16" in name or "wino_conv3x3_split" in name:
        cur.append(us)
    elif "prep" in name and cur:
        groups.append(cur); cur = []
if cur: groups.append(cur)
print(" ".join("%.1f" % statistics.median(g) for g in groups), "| sum %.1f" % sum(statistics.median(g) for g in groups))
