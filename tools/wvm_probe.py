"""One five-stage detector of ffpDetectApp/*.cfg alone on a 1080p frame: per-call wall time and the hipEvent-bracketed time of its
WVM kernels.  Run under `rocprofv3 --kernel-trace --stats` / `--pmc ...` for per-kernel durations and counters in isolation.
usage: wvm_probe.py [detector name] [iterations]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from featuredetection_amd import capi, synth

name = sys.argv[1] if len(sys.argv) > 1 else "LeftLipCorner"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctx = capi.Context(0)
m = [x for x in bench.ffp15_models(nsv=256) if x[0] == name][0]
_, key, wm, sm, pw, ph = m
frame = synth.make_frame(1920, 1080, seed=20260927)
p = capi.Pyramid(ctx, inc=float(np.float32(key[0])), min_scale=float(np.float32(key[1])), max_scale=float(np.float32(key[2])))
p.update(frame)
w, s = capi.Wvm(ctx, wm), capi.Svm(ctx, sm)
n = p.window_count(pw, ph, 1, 1)
ctx.set_kernel_timing(True)
ts, ks = [], []
for i in range(iters):
    t0 = time.perf_counter()
    d, st = capi.detect_five_stage(ctx, p, w, s, cap=1 << 14)
    ts.append(time.perf_counter() - t0)
    ks.append(ctx.last_kernel_ms()[1])
print("%s %dx%d: %d windows, stages %s, call %.3f ms, WVM kernels %.3f ms (%.1f Mwindows/s kernel-only)" %
      (name, pw, ph, n, st.tolist(), 1e3 * np.median(ts[2:]), np.median(ks[2:]), n / np.median(ks[2:]) / 1e3))
