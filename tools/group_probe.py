"""Dev tool: the detectors of ffpDetectApp that share one pyramid and patch size (default: the seven 24x24 ones) as one five-stage
batch on a 1080p frame -- wall time per batch with separate pre-filters (default) and with the shared one (FD_WVM_GROUP=1)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
import bench
from featuredetection_amd import capi, synth

size = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "24x24").split("x"))
ctx = capi.Context(0)
models = [m for m in bench.ffp15_models(nsv=256) if (m[4], m[5]) == size and tuple(np.float32(m[1])) == tuple(np.float32((0.9, 0.5, 0.7)))] or \
         [m for m in bench.ffp15_models(nsv=256) if (m[4], m[5]) == size]
key = models[0][1]
models = [m for m in models if m[1] == key]
frame = synth.make_frame(1920, 1080, seed=20260927)
p = capi.Pyramid(ctx, inc=float(np.float32(key[0])), min_scale=float(np.float32(key[1])), max_scale=float(np.float32(key[2])))
p.update(frame)
dets = [(p, capi.Wvm(ctx, m[2]), capi.Svm(ctx, m[3])) for m in models]
ts = []
for i in range(8):
    t0 = time.perf_counter()
    res = capi.FiveStageBatch(ctx, dets, cap=1 << 14).end()
    ts.append(time.perf_counter() - t0)
n = p.window_count(size[0], size[1], 1, 1)
print("%d detectors %dx%d, %d windows each, FD_WVM_GROUP=%s: %.3f ms per batch, detections %s" %
      (len(dets), size[0], size[1], n, os.environ.get("FD_WVM_GROUP", "0"), 1e3 * float(np.median(ts[2:])), [len(d) for d, _ in res]))
