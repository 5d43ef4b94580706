"""Forward / backward-data / weight-gradient time of the Winograd kernels with every tools/micro/libdcd_*.so (ablation builds of
conv.hip) in place of the product library: one line per library."""
import glob, os, shutil, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, R)
    import torch
    from dcd_amd import ops
    dev = torch.device("cuda:0")
    out = []
    for C, K, H, W in [(64, 64, 96, 320), (128, 128, 48, 160), (256, 256, 24, 80), (512, 512, 12, 40)]:
        x = torch.randn(8, C, H, W, device=dev)
        w = torch.randn(K, C, 3, 3, device=dev) * 0.05
        gy = torch.randn(8, K, H, W, device=dev)
        tf, tb = ops.conv3x3_transform_weights(w)

        def t(fn, n=20):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n * 1e3
        out.append("%d@%d: f %.0f w %.0f us" % (C, H, t(lambda: ops._conv3x3_call(x, w, K, False, transformed=tf)),
                                               t(lambda: ops._conv3x3_wrw_call(x, gy, w.shape))))
    print(" | ".join(out))
    sys.exit(0)
main = os.path.join(R, "dcd_amd", "libdcd_hip.so")
shutil.copy(main, main + ".bak")
try:
    for f in [main + ".bak"] + sorted(glob.glob(os.path.join(R, "tools", "micro", "libdcd_*.so"))):
        shutil.copy(f, main)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], capture_output=True, text=True)
        print("%-26s %s" % (os.path.basename(f), (out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1]))
finally:
    shutil.copy(main + ".bak", main)
