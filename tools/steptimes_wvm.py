import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import torch
from featuredetection_amd import capi, synth
from oracle import pyoracle as O
W, H = [int(v) for v in sys.argv[1].split('x')]
ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
frames = [synth.make_frame(W, H, seed=20260927 + i) for i in range(2)]
d = [torch.from_numpy(f).cuda() for f in frames]
gray = O.bgr2gray(synth.make_frame(640, 480, seed=20260927))
calib = synth.random_patches(gray[::4, ::4].copy(), 20, 20, 8000, np.random.default_rng(1))
wvm_m = synth.make_wvm(7, calib_patches=calib)
import os
if os.environ.get("NUM_USED"): wvm_m["num_used"] = int(os.environ["NUM_USED"])
eq = synth.histeq64_np(synth.random_patches(gray[::4, ::4].copy(), 20, 20, 1400, np.random.default_rng(2)))
svm_m = synth.make_svm_u8(3, eq, nsv=1024, calib=eq[1024:])
pyr = capi.Pyramid(ctx, inc=float(np.float32(0.92)), min_scale=float(np.float32(0.05)), max_scale=float(np.float32(0.16)))
wvm, svm = capi.Wvm(ctx, wvm_m), capi.Svm(ctx, svm_m)
ts = []
for i in range(30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pyr.update_device(d[i % 2].data_ptr(), W, H, 3)
    t1 = time.perf_counter()
    dets, st = capi.detect_five_stage(ctx, pyr, wvm, svm)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, ctx.last_kernel_ms()[1], st.tolist()))
for t in ts[::3]: print("update %.3f five %.3f kernel %.3f stages %s" % t)
