"""Dev tool: wall time of the host calls of the headline workload (multi-frame five-stage cascade, two calls in flight)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from featuredetection_amd import capi, synth

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 32
NSLOT = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
frames = [torch.from_numpy(synth.make_frame(640, 480, seed=20260927 + i)).to(dev) for i in range(8)]
wm, sm = bench.cascade_models()
kw = dict(inc=float(np.float32(0.92)), min_scale=float(np.float32(0.05)), max_scale=float(np.float32(0.16)))
slots = []
for k in range(NSLOT):
    c = capi.Context(0)
    p = capi.Pyramid(c, **kw); p.set_frames(NB)
    slots.append(dict(ctx=c, pyr=p, wvm=capi.Wvm(c, wm), svm=capi.Svm(c, sm), run=None))
T = dict(update=0.0, begin=0.0, end=0.0)
ncall = 0
def call(i, timed):
    global ncall
    sl = slots[i % NSLOT]
    t0 = time.perf_counter()
    if sl["run"] is not None:
        sl["run"].end(); sl["run"] = None
    t1 = time.perf_counter()
    sl["pyr"].update_frames(device_ptrs=[frames[(i * NB + j) % 8].data_ptr() for j in range(NB)], w=640, h=480, ch=3)
    t2 = time.perf_counter()
    sl["run"] = capi.FiveStageFrames(sl["ctx"], sl["pyr"], sl["wvm"], sl["svm"], NB)
    t3 = time.perf_counter()
    if timed:
        T["end"] += t1 - t0; T["update"] += t2 - t1; T["begin"] += t3 - t2; ncall += 1
for i in range(8): call(i, False)
t0 = time.perf_counter()
N = 64
for i in range(8, 8 + N): call(i, True)
for sl in slots:
    if sl["run"] is not None: sl["run"].end(); sl["run"] = None
wall = time.perf_counter() - t0
print("frames/call %d, %d slots: %.3f ms per call wall; host: end %.3f  update_frames %.3f  begin %.3f ms per call" %
      (NB, NSLOT, 1e3 * wall / N, 1e3 * T["end"] / ncall, 1e3 * T["update"] / ncall, 1e3 * T["begin"] / ncall))
