"""Dev tool: how many wavefronts of k_wvm_prefilter are resident per SIMD over the launch, and where a tile's time goes, from per-wavefront
records (start, end, HW_ID, phase ticks) of a -DFD_WVB_PROF build (FD_HIP_LIB=featuredetection_amd/alt/libfd_hip_wvbprof.so).
usage: wvd_residency.py [nframes]"""
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
L.fd_debug_wvd_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.fd_debug_wvd_prof.restype = ctypes.c_int
cap = L.fd_debug_wvd_prof(None, 0)
ctx = capi.Context(0)
wm, sm = bench.cascade_models()
frames = [synth.make_frame(640, 480, seed=20260927 + i) for i in range(8)]
p = capi.Pyramid(ctx, inc=float(np.float32(0.92)), min_scale=float(np.float32(0.05)), max_scale=float(np.float32(0.16)))
p.set_frames(NB)
w, s = capi.Wvm(ctx, wm), capi.Svm(ctx, sm)
def run():
    p.update_frames(images=[frames[j % 8] for j in range(NB)])
    return capi.detect_five_stage_frames(ctx, p, w, s, NB)
for _ in range(3):
    run()
ctx.synchronize()
rec = np.zeros((cap, 8), dtype=np.uint64)
L.fd_debug_wvd_prof(rec.ctypes.data_as(ctypes.c_void_p), cap)
rec = rec[rec[:, 1] > 0]
T = 1.0 / 2400.0   # us per tick if s_memtime runs at the shader clock; the spans below are also given relative to the launch
t0, t1 = rec[:, 0].astype(np.int64), rec[:, 1].astype(np.int64)
hw = (rec[:, 2] & 0xffffffff).astype(np.int64)
xcc = (rec[:, 2] >> 32).astype(np.int64) & 0xf
tiles = rec[:, 3].astype(np.int64)
# the counters of different XCDs / shader engines are not aligned: times relative to the first wavefront of the same CU
cuKey = ((xcc * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 16 + ((hw >> 8) & 0xf)
for x in np.unique(cuKey):
    m = cuKey == x
    base = t0[m].min()
    t0[m] -= base; t1[m] -= base
span = int(t1.max() - t0.min())
print("CUs seen %d; per-CU span us: min %.1f mean %.1f max %.1f" % (len(np.unique(cuKey)), min((t1[cuKey == x].max()) for x in np.unique(cuKey)) * T,
      np.mean([t1[cuKey == x].max() for x in np.unique(cuKey)]) * T, span * T))
print("waves %d, tiles %d (per wave min %d max %d), launch span %d ticks = %.1f us at 2.4 GHz" % (len(rec), tiles.sum(), tiles.min(), tiles.max(), span, span * T))
simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = ((xcc * 8 + se) * 2 + sh) * 64 + cu * 4 + simd
uk = np.unique(key)
print("distinct SIMDs seen %d; distinct XCC %d SE %d CU ids %d" % (len(uk), len(np.unique(xcc)), len(np.unique(se)), len(np.unique(cu))))
life = (t1 - t0).astype(np.float64)
print("wave lifetime us: mean %.1f min %.1f max %.1f; sum of lifetimes / (SIMDs x span) = %.2f resident wavefronts per SIMD on average"
      % (life.mean() * T, life.min() * T, life.max() * T, life.sum() / (len(uk) * span)))
# residency histogram over time, all SIMDs together
edges = np.linspace(t0.min(), t1.max(), 41)
for a, b in zip(edges[:-1], edges[1:]):
    ov = np.clip(np.minimum(t1, b) - np.maximum(t0, a), 0, None).sum() / ((b - a) * len(uk))
    print("  t %6.1f us: %.2f" % ((a - t0.min()) * T, ov))
ph = rec[:, 4:8].astype(np.float64).sum(axis=0) / max(tiles.sum(), 1) * T
print("per tile us: histogram %.2f, cdf+LUT %.2f, equalise+MFMA %.2f, transposes+levels+queue %.2f; sum %.2f; wave lifetime per tile %.2f"
      % (ph[0], ph[1], ph[2], ph[3], ph.sum(), life.sum() * T / tiles.sum()))
st = (t0 - t0.min()) * T
print("wave start times us: p10 %.1f p50 %.1f p90 %.1f max %.1f" % tuple(np.percentile(st, [10, 50, 90, 100])))
for n in np.unique(tiles):
    m = tiles == n
    print("  waves with %d tiles: %d, lifetime mean %.1f us" % (n, m.sum(), life[m].mean() * T))
