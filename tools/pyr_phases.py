"""Dev tool: where a k_resize_down workgroup spends its time (thread 0's s_memtime ticks per phase), from a -DFD_PYR_PROF build
(FD_HIP_LIB=featuredetection_amd/alt/libfd_hip_wvbprof.so).  usage: pyr_phases.py [nframes]"""
import ctypes, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401  (before libfd_hip.so)
from featuredetection_amd import capi, synth

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = capi.lib()
L.fd_debug_pyr_prof.argtypes = [ctypes.c_void_p]
L.fd_debug_pyr_prof.restype = ctypes.c_int
cap = L.fd_debug_pyr_prof(None)
ctx = capi.Context(0)
frames = [synth.make_frame(640, 480, seed=20260927 + i) for i in range(8)]
p = capi.Pyramid(ctx, inc=float(np.float32(0.92)), min_scale=float(np.float32(0.05)), max_scale=float(np.float32(0.16)))
p.set_frames(NB)
def run():
    p.update_frames(images=[frames[j % 8] for j in range(NB)])
    ctx.synchronize()
for _ in range(3):
    run()
rec = np.zeros((cap, 8), dtype=np.uint64)
L.fd_debug_pyr_prof(rec.ctypes.data_as(ctypes.c_void_p))
rec = rec[rec[:, 5] > 0].astype(np.float64)
work = rec[rec[:, :5].sum(axis=1) > 0]
T = 1.0 / 2400
names = ["stage source + row table (to the barrier)", "resize", "wait at the barrier", "pyrDown + store", "kept-layer copy + barrier"]
tot = work[:, :5].sum()
print("%d workgroups, %d with a tile; per working workgroup us:" % (len(rec), len(work)))
for i, n in enumerate(names):
    print("  %-44s %7.2f  (%4.1f %%)" % (n, work[:, i].mean() * T, 100.0 * work[:, i].sum() / max(tot, 1)))
print("  total %.2f us per workgroup" % (tot * T / max(len(work), 1)))
