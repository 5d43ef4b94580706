import csv, glob, sys
for f in sorted(glob.glob('/root/repo/gpurun_out/r02_dense_*_d%s.csv' % (sys.argv[1] if len(sys.argv) > 1 else '?'))):
    rows = list(csv.DictReader(open(f)))
    print('==', f.split('r02_dense_')[1])
    tot = 0
    for r in rows:
        k = r['kernel']
        if 'at::native' in k:
            continue
        tot += float(r['ms_per_step'])
        if float(r['ms_per_step']) < 0.02:
            continue
        k = k.replace('(anonymous namespace)::', '').replace('void ', '')
        print(f"  {float(r['avg_us']):8.1f} us x{float(r['calls_per_step']):.0f}  {k[:60]}")
    print('  total ms per call', round(tot / 4, 3))
