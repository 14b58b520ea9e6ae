import os, sys, ctypes as C
sys.argv = ["x"]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from featuredetection_amd import capi
env = bench.Env(); env.world = 1; env.rank = 0; env.local_rank = 0
torch.cuda.set_device(0); env.dev = torch.device("cuda", 0)
env.ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream); env.dist = None
wl = bench.Ffp15(env)
sl = wl.slots[0]
os.environ["FD_FS_TAIL"] = "1"
L = capi.lib()
for fi in (0, 5):
    fr = wl.dframes[fi]
    for di in (0, 4, 6, 9):
        name, pr, wv, sv_, pw, ph = sl["dets"][di]
        pr.update_device(fr.data_ptr(), wl.W, wl.H, 3)
        for rep in range(2):
            r = capi.FiveStageBatch(env.ctx, [(pr, wv, sv_)], cap=4096).end()
        buf = (C.c_ulonglong * 8)()
        L.fd_debug_fst_prof(buf)
        t = list(buf)
        print(name, "n", t[5], "keep", t[6], "us: keys+zero->sort %.1f geo %.1f sweep %.1f out %.1f" % ((t[1]-t[0])/100, (t[2]-t[1])/100, (t[3]-t[2])/100, (t[4]-t[3])/100), "state", wv.last_tail_state())
