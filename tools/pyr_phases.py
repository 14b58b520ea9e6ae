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
rec = rec[rec[:, 6] > 0].astype(np.float64)
T = 1.0 / 2400
names = ["fetched tile -> LDS (waits for its loads)", "barrier", "issue the next tile's loads", "resize", "barrier", "pyrDown + stores"]
tiles = rec[:, 6].sum()
tot = rec[:, :6].sum()
print("%d workgroups, %d tiles; per tile us (thread 0):" % (len(rec), int(tiles)))
for i, n in enumerate(names):
    print("  %-44s %7.2f  (%4.1f %%)" % (n, rec[:, i].sum() * T / tiles, 100.0 * rec[:, i].sum() / max(tot, 1)))
print("  total %.2f us per tile" % (tot * T / tiles))
