"""Dev tool: phase split of k_sdm_descriptors from in-kernel timestamps (a libfd_hip.so built with -DFD_SDM_PROF, FD_HIP_LIB=...)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
import bench
from featuredetection_amd import capi

class Env: pass
env = Env(); env.rank = 0; env.world = 1; env.local_rank = 0; env.dev = torch.device("cuda:0"); env.ctx = capi.Context(0)
os.environ["FD_BENCH_SDM_THREADS"] = "1"
wl = bench.Sdm(env, batches_per_step=2)
L = capi.lib()
L.fd_debug_sdm_prof.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
buf = (ctypes.c_ulonglong * 8)()
wl.step(0)
L.fd_debug_sdm_prof(buf, 1)
wl.step(1); wl.step(2)
L.fd_debug_sdm_prof(buf, 0)
v = [int(x) for x in buf]
names = ["items", "crop + resize", "gradients + masks", "voting", "norms + assembly", "transpose / store"]
print("items %d, cycles per item %.0f" % (v[0], v[6] / max(v[0], 1)))
for i in range(1, 6):
    print("  %-20s %5.1f %%  %8.0f cycles/item" % (names[i], 100.0 * v[i] / v[6], v[i] / max(v[0], 1)))
