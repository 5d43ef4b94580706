"""Per-kernel times of ONE DCN layer for every tools/micro/libdcd_*.so variant (built by build_variants.sh with ablation macros).
usage: prof_layer_variants.py C Co H W [kernel-substring ...]"""
import csv, glob, os, shutil, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
main = os.path.join(R, "dcd_amd", "libdcd_hip.so")
args = sys.argv[1:5]
want = sys.argv[5:] or ["dcn_"]
shutil.copy(main, main + ".bak")
try:
    for f in sorted(glob.glob(os.path.join(R, "tools", "micro", "libdcd_*.so"))):
        shutil.copy(f, main)
        d = "/tmp/plv_" + os.path.basename(f)
        shutil.rmtree(d, ignore_errors=True)
        subprocess.run("cd /tmp && timeout 300 rocprofv3 --kernel-trace -d %s -- python3 %s/tools/one_layer.py %s > /dev/null 2>&1" % (
            d, R, " ".join(args)), shell=True)
        dbs = glob.glob(d + "/**/*.db", recursive=True)
        if not dbs:
            print(os.path.basename(f), "no trace")
            continue
        subprocess.run([sys.executable, os.path.join(R, "tools", "prof_summary.py"), os.path.dirname(dbs[0]), d + ".csv"], capture_output=True)
        print(os.path.basename(f))
        for r in list(csv.reader(open(d + ".csv")))[1:]:
            if any(w in r[0] for w in want):
                print("   %-60s %8s us" % (r[0][28:88], r[3]))
finally:
    shutil.copy(main + ".bak", main)
