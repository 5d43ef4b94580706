import sys, os, shutil, subprocess, glob
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
main = os.path.join(R, "dcd_amd", "libdcd_hip.so")
shutil.copy(main, main + ".bak")
try:
    for f in sorted(glob.glob(os.path.join(R, "tools", "micro", "libdcd_*.so"))):
        shutil.copy(f, main)
        d = "/tmp/pv_" + os.path.basename(f)
        subprocess.run("cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d %s -o x -- python %s/tools/dcn_one_pass.py 8 0.5 > /dev/null 2>&1" % (d, R), shell=True)
        out = subprocess.run([sys.executable, os.path.join(R, "tools", "prof_summary.py"), d, d + ".csv"], capture_output=True, text=True).stdout
        print(os.path.basename(f), out.strip())
        import csv
        for r in list(csv.reader(open(d + ".csv")))[1:9]:
            print("   %-52s %8s ms" % (r[0][28:80], r[2]))
finally:
    shutil.copy(main + ".bak", main)
