import os, sys
sys.argv = ["x"]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from featuredetection_amd import capi
env = bench.Env(); env.world = 1; env.rank = 0; env.local_rank = 0
torch.cuda.set_device(0); env.dev = torch.device("cuda", 0)
env.ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream); env.dist = None
wl = bench.Ffp15(env)
sl = wl.slots[0]
bad = 0
for fi in range(0, 32, 3):
    fr = wl.dframes[fi]
    res = {}
    for tail in ("0", "1"):
        os.environ["FD_FS_TAIL"] = tail
        for pr in sl["pyrs"].values():
            pr.update_device(fr.data_ptr(), wl.W, wl.H, 3)
        r = capi.FiveStageBatch(env.ctx, [(pr, wv, sv_) for _, pr, wv, sv_, _, _ in sl["dets"]], cap=4096).end()
        res[tail] = (r, [d[2].last_tail_state() for d in sl["dets"]])
    for di, ((d0, s0), (d1, s1)) in enumerate(zip(res["0"][0], res["1"][0])):
        same = d0.tobytes() == d1.tobytes() and np.array_equal(s0, s1)
        if not same:
            bad += 1
            print("frame", fi, "det", di, sl["dets"][di][0], "host", s0.tolist(), "dev", s1.tolist(), "state", res["1"][1][di])
print("states of last frame", res["1"][1], "mismatching jobs", bad)
