"""Times one layer's backward with every tools/micro/libdcd_*.so in place of the product library (ablation builds)."""
import sys, os, shutil, subprocess, glob
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
main = os.path.join(R, "dcd_amd", "libdcd_hip.so")
shutil.copy(main, main + ".bak")
args = sys.argv[1:] or ["64", "64", "96", "320"]
try:
    for f in sorted(glob.glob(os.path.join(R, "tools", "micro", "libdcd_*.so"))):
        shutil.copy(f, main)
        out = subprocess.run([sys.executable, os.path.join(R, "tools", "time_bwd_one.py")] + args, capture_output=True, text=True)
        print("%-28s %s" % (os.path.basename(f), (out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1]))
finally:
    shutil.copy(main + ".bak", main)
