"""Dev tool: phase split of k_wvm_deepB (stage B of the WVM cascade) from in-kernel timestamps.  Needs a libfd_hip.so built with -DFD_DEEP4_PROF
(FD_HIP_LIB=.../libfd_hip_prof.so); timestamps are taken by thread 0 of each workgroup (wave 0), s_memtime ticks (100 MHz).
usage: deep4_phases.py [cascade|<ffp15 detector name>]"""
import ctypes, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401  (before libfd_hip.so)
import bench
from featuredetection_amd import capi, synth

what = sys.argv[1] if len(sys.argv) > 1 else "cascade"
L = capi.lib()
L.fd_debug_deep4_prof.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
buf = (ctypes.c_ulonglong * 16)()
ctx = capi.Context(0)
if what == "cascade":
    wm, sm = bench.cascade_models()
    NB = 32
    frames = [synth.make_frame(640, 480, seed=20260927 + i) for i in range(8)]
    p = capi.Pyramid(ctx, inc=float(np.float32(0.92)), min_scale=float(np.float32(0.05)), max_scale=float(np.float32(0.16)))
    p.set_frames(NB)
    w, s = capi.Wvm(ctx, wm), capi.Svm(ctx, sm)
    def run():
        p.update_frames(images=[frames[j % 8] for j in range(NB)])
        return capi.detect_five_stage_frames(ctx, p, w, s, NB)
else:
    _, key, wm, sm, pw, ph = [x for x in bench.ffp15_models(nsv=256) if x[0] == what][0]
    frame = synth.make_frame(1920, 1080, seed=20260927)
    p = capi.Pyramid(ctx, inc=float(np.float32(key[0])), min_scale=float(np.float32(key[1])), max_scale=float(np.float32(key[2])))
    p.update(frame)
    w, s = capi.Wvm(ctx, wm), capi.Svm(ctx, sm)
    def run():
        return capi.detect_five_stage(ctx, p, w, s, cap=1 << 14)
run(); run()
L.fd_debug_deep4_prof(buf, 1)
N = 5
for _ in range(N):
    res = run()
L.fd_debug_deep4_prof(buf, 0)
v = [int(x) for x in buf]
names = ["windows", "prepare", "chunks", "kernel values", "barrier 1", "hier sums", "barrier 2", "emit", "total", "levels evaluated"]
tot = v[8]
print("%s: per launch %d batches of 4 windows, %d chunks, mean levels (window of wave 0) %.1f" % (what, v[0] // N, v[2] // N, v[9] / max(v[0], 1)))
for i in (1, 3, 4, 5, 6, 7):
    print("  %-14s %5.1f %%   %.0f ticks/batch" % (names[i], 100.0 * v[i] / tot, v[i] / max(v[0], 1)))
print("  total ticks/batch %.0f" % (tot / max(v[0], 1)))
if v[12]:
    print("  inside kernel values: rect passes %.0f ticks/generation, chain + exp %.0f ticks/generation (%d generations/batch)" %
          (v[10] / v[12], v[11] / v[12], v[12] // max(v[0], 1)))
