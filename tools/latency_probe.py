"""Single-frame latency of the headline detector: one 640x480 BGR frame resident in HBM -> detections on the host, one frame at a
time (fd_pyramid_update + fd_detect_five_stage, blocking; and the same through the one-call entry point fd_detect_five_stage_image).  Prints p50 / p90 / p99 over N frames and the stage split from FD_TRACE-free
host timers."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from featuredetection_amd import capi, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
wm, sm = bench.cascade_models()
kw = dict(inc=float(np.float32(0.92)), min_scale=float(np.float32(0.05)), max_scale=float(np.float32(0.16)))
pyr = capi.Pyramid(ctx, **kw)
w, s = capi.Wvm(ctx, wm), capi.Svm(ctx, sm)
frames = [torch.from_numpy(synth.make_frame(640, 480, seed=20260927 + i)).cuda() for i in range(8)]
torch.cuda.synchronize()
def one(i):
    f = frames[i % 8]
    pyr.update_device(f.data_ptr(), 640, 480, 3)
    return capi.detect_five_stage(ctx, pyr, w, s)
for i in range(50):
    one(i)
lat, tu = [], []
for i in range(N):
    t0 = time.perf_counter()
    f = frames[i % 8]
    pyr.update_device(f.data_ptr(), 640, 480, 3)
    t1 = time.perf_counter()
    d, st = capi.detect_five_stage(ctx, pyr, w, s)
    t2 = time.perf_counter()
    lat.append((t2 - t0) * 1e6)
    tu.append((t1 - t0) * 1e6)
lat = np.array(lat); tu = np.array(tu)
one_call = capi.FiveStageImage(ctx, pyr, w, s)
lat2 = []
for i in range(50):
    one_call.detect_device(frames[i % 8].data_ptr(), 640, 480, 3)
for i in range(N):
    t0 = time.perf_counter()
    d2, st2 = one_call.detect_device(frames[i % 8].data_ptr(), 640, 480, 3)
    lat2.append((time.perf_counter() - t0) * 1e6)
lat2 = np.array(lat2)
print("fd_detect_five_stage_image (Detector::detect(image): update + detect in one call) us: p50 %.1f p90 %.1f p99 %.1f mean %.1f"
      % (np.percentile(lat2, 50), np.percentile(lat2, 90), np.percentile(lat2, 99), lat2.mean()))
print("single-frame latency us: p50 %.1f p90 %.1f p99 %.1f mean %.1f (pyramid-update call returns after %.1f us); detections/frame %d; %.1f Mpatches/s at batch 1"
      % (np.percentile(lat, 50), np.percentile(lat, 90), np.percentile(lat, 99), lat.mean(), np.median(tu), len(d), 16185 / np.percentile(lat, 50)))
