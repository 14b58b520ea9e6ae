"""Where the bench thread's time goes in the ffp15 workload: seconds inside end / update / begin / the rest (python) per frame."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from featuredetection_amd import capi

os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
env = bench.Env()
env.world, env.rank, env.local_rank = 1, 0, 0
env.dev = torch.device("cuda:0")
torch.cuda.set_device(0)
env.ctx = capi.Context(0)
wl = bench.Ffp15(env)
T = dict(end=0.0, upd=0.0, begin=0.0)
o_end, o_init, o_upd = capi.FiveStageBatch.end, capi.FiveStageBatch.__init__, capi.Pyramid.update_device
def t_end(self):
    t = time.perf_counter(); r = o_end(self); T["end"] += time.perf_counter() - t; return r
def t_init(self, *a, **k):
    t = time.perf_counter(); o_init(self, *a, **k); T["begin"] += time.perf_counter() - t
def t_upd(self, *a, **k):
    t = time.perf_counter(); r = o_upd(self, *a, **k); T["upd"] += time.perf_counter() - t; return r
capi.FiveStageBatch.end, capi.FiveStageBatch.__init__, capi.Pyramid.update_device = t_end, t_init, t_upd
for i in range(3):
    wl.step(i)
wl.flush(); torch.cuda.synchronize()
for k in T: T[k] = 0.0
t0 = time.perf_counter()
N = 10
for i in range(N):
    wl.step(i)
wl.flush(); torch.cuda.synchronize()
dt = time.perf_counter() - t0
nf = N * wl.FP
print("per frame ms: total %.3f end %.3f update %.3f begin(+marshal) %.3f other %.3f" % (1e3 * dt / nf, 1e3 * T["end"] / nf, 1e3 * T["upd"] / nf, 1e3 * T["begin"] / nf, 1e3 * (dt - sum(T.values())) / nf))
