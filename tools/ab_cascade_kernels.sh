# isolated duration of the cascade kernels (pre-filter + stage B) of one multi-frame call, per variant / environment setting
# usage: tools/ab_cascade_kernels.sh "VAR=value ..." ["VAR=value ..."]   (an empty string = defaults); FD_HIP_LIB selects another build
R=$PWD
for rep in 1 2; do for envs in "$@"; do
  env $envs python - <<PY
import os, sys
sys.path.insert(0, "$R")
os.environ.setdefault("GPU_MAX_HW_QUEUES","8")
import numpy as np, torch, bench
from featuredetection_amd import capi, synth
ctx = capi.Context(0)
wm, sm = bench.cascade_models()
NB = 64
frames = [synth.make_frame(640, 480, seed=20260927 + i) for i in range(8)]
p = capi.Pyramid(ctx, inc=float(np.float32(0.92)), min_scale=float(np.float32(0.05)), max_scale=float(np.float32(0.16)))
p.set_frames(NB)
w, s = capi.Wvm(ctx, wm), capi.Svm(ctx, sm)
ctx.set_kernel_timing(True)
ms = []
for i in range(14):
    p.update_frames(images=[frames[(i + j) % 8] for j in range(NB)])
    capi.detect_five_stage_frames(ctx, p, w, s, NB)
    ms.append(ctx.last_kernel_ms()[1])
print("[$envs]: cascade kernels %.4f ms (min %.4f)" % (float(np.mean(ms[3:])), float(np.min(ms[3:]))))
PY
done; done
