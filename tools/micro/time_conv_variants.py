import sys, os, shutil, subprocess, glob
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
main = os.path.join(R, "dcd_amd", "libdcd_hip.so")
shutil.copy(main, main + ".bak")
try:
    for f in sorted(glob.glob(os.path.join(R, "tools", "micro", "libdcd_*.so"))):
        shutil.copy(f, main)
        out = subprocess.run([sys.executable, os.path.join(R, "tools", "time_conv.py")], capture_output=True, text=True).stdout
        print(os.path.basename(f))
        print("\n".join(l for l in out.splitlines()[:4]))
finally:
    shutil.copy(main + ".bak", main)
