import sys, numpy as np
sys.path.insert(0, '/root/repo')
import torch
from featuredetection_amd import capi, synth
from oracle import pyoracle as O
W, H = [int(v) for v in sys.argv[1].split('x')]
ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
frame = synth.make_frame(W, H, seed=20260927)
gray = O.bgr2gray(synth.make_frame(640, 480, seed=20260927))
calib = synth.random_patches(gray[::4, ::4].copy(), 20, 20, 8000, np.random.default_rng(1))
wvm_m = synth.make_wvm(7, calib_patches=calib)
print("filters", wvm_m['num_filters'] if isinstance(wvm_m, dict) else getattr(wvm_m, 'num_filters', None))
pyr = capi.Pyramid(ctx, inc=float(np.float32(0.92)), min_scale=float(np.float32(0.05)), max_scale=float(np.float32(0.16)))
wvm = capi.Wvm(ctx, wvm_m)
pyr.update(frame)
dets, lev, sc = capi.detect_wvm(ctx, pyr, wvm, want_all=True)
lev = np.asarray(lev)
print("windows", lev.size, "mean exit level", lev.mean() + 1, "sum levels", (lev + 1).sum())
for t in (1, 2, 4, 8, 16, 32, 64, 128, 256, 279):
    print("reach level >=", t, (lev >= t).sum(), "share of level evals beyond", ((lev + 1 - t).clip(0)).sum() / (lev + 1).sum())
