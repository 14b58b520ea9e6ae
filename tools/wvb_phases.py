"""Dev tool: phase split of the dense stage-B kernels (wvm_stageb.hpp) from in-kernel timestamps.  Needs a libfd_hip.so built with
-DFD_WVB_PROF (FD_HIP_LIB=featuredetection_amd/alt/libfd_hip_wvbprof.so); thread 0 of every workgroup stamps, s_memtime ticks (100 MHz).
usage: wvb_phases.py [nframes]"""
import ctypes, os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401  (before libfd_hip.so)
import bench
from featuredetection_amd import capi, synth

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = capi.lib()
L.fd_debug_wvb_prof.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
buf = (ctypes.c_ulonglong * 64)()
ctx = capi.Context(0)
wm, sm = bench.cascade_models()
frames = [synth.make_frame(640, 480, seed=20260927 + i) for i in range(8)]
p = capi.Pyramid(ctx, inc=float(np.float32(0.92)), min_scale=float(np.float32(0.05)), max_scale=float(np.float32(0.16)))
p.set_frames(NB)
w, s = capi.Wvm(ctx, wm), capi.Svm(ctx, sm)
def run():
    p.update_frames(images=[frames[j % 8] for j in range(NB)])
    return capi.detect_five_stage_frames(ctx, p, w, s, NB)
run(); run()
L.fd_debug_wvb_prof(buf, 1)
N = 5
for _ in range(N):
    run()
L.fd_debug_wvb_prof(buf, 0)
v = [int(x) for x in buf]
T = 1.0 / 2400   # us per tick (shader clock, ~2.4 GHz at most)
for ph in range(3):
    c = v[8 * ph:8 * ph + 8]
    if c[0]:
        print("chain phase %d: %d units/launch; per unit (thread 0 = wave 0): stage X %.2f us, records %.2f, GEMM %.2f, chain %.2f (%.1f levels, %.2f us/level), total %.2f us"
              % (ph, c[0] // N, c[1] * T / c[0], c[2] * T / c[0], c[3] * T / c[0], c[4] * T / c[0], c[6] / c[0], c[4] * T / max(c[6], 1), c[5] * T / c[0]))
        if c[7]:
            print("   of stage X: %.2f us until the windows' pixels are in LDS (behind the first tile's fragments and records in vmcnt order)" % (c[7] * T / c[0]))
for ph in range(3):
    c = v[24 + 4 * ph:24 + 4 * ph + 4]
    if c[0]:
        print("sums phase %d: %d units/launch; weights %.2f us, terms %.2f us per unit" % (ph, c[0] // N, c[1] * T / c[0], c[2] * T / c[0]))
for ph in range(3):
    c = v[36 + 8 * ph:36 + 8 * ph + 8]
    if c[6]:
        print("exit phase %d: %d tiles, %d workgroups/launch; per tile: key+alloc %.2f us (loads %.2f, atomics %.2f), positives %.2f, survivors %.2f; per workgroup: total %.2f us"
              % (ph, c[0] // N, c[6] // N, c[1] * T / max(c[0], 1), c[4] * T / max(c[0], 1), c[7] * T / max(c[0], 1), c[2] * T / max(c[0], 1), c[3] * T / max(c[0], 1), c[5] * T / c[6]))
