"""N>1 path on CPU: two gloo ranks shard a batch of images (rank r takes images r, r+2, ...), run a
CPU stand-in for the per-image detector (the oracle) and exchange ONE gather of detection records.
The gathered, ordered record list must equal the single-process result."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _detect_image(i):
    """deterministic fake detector: image id -> structured detections"""
    from featuredetection_amd import capi
    rng = np.random.default_rng(1000 + i)
    n = int(rng.integers(0, 6))
    d = np.zeros(n, capi.DET_DTYPE)
    d["cx"], d["cy"] = rng.integers(0, 640, n), rng.integers(0, 480, n)
    d["w"] = d["h"] = rng.integers(80, 200, n)
    d["score"], d["probability"] = rng.random(n), rng.random(n)
    return d


def _worker(rank, world, port, nimages, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from featuredetection_amd import parallel
    mine = parallel.shard_indices(nimages, rank, world)
    recs = [parallel.pack_records(np.full(len(d), i), np.zeros(len(d)), d) for i in mine for d in [_detect_image(i)]]
    local = np.concatenate(recs) if recs else np.zeros((0, parallel.RECORD_FIELDS))
    allr, trunc = parallel.gather_records(local, cap=256)
    if rank == 0:
        q.put((allr, trunc))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_gather_world2():
    from featuredetection_amd import parallel
    nimages, world = 11, 2
    assert parallel.shard_indices(nimages, 0, 2) == [0, 2, 4, 6, 8, 10]
    assert parallel.shard_indices(nimages, 1, 2) == [1, 3, 5, 7, 9]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nimages, q)) for r in range(world)]
    for p in procs:
        p.start()
    allr, trunc = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert not trunc
    ref = [parallel.pack_records(np.full(len(d), i), np.zeros(len(d)), d) for i in range(nimages) for d in [_detect_image(i)]]
    ref = np.concatenate(ref)
    assert allr.shape == ref.shape
    assert np.array_equal(allr, ref)   # ordered by image id, original order inside an image


def _golden_image_detections(i, nimages):
    """fd_detection records of "image" i: a slice of the committed cascade fixture's WVM positives (tests/golden/
    orc_cascade_160x120.npz, written by the oracle), converted to the C ABI's record layout"""
    from featuredetection_amd import capi
    g = np.load(os.path.join(ROOT, "tests", "golden", "orc_cascade_160x120.npz"))
    src = g["wvm_pos"][i::nimages]
    d = np.zeros(len(src), capi.DET_DTYPE)
    for f in ("cx", "cy", "w", "h", "layer", "lx", "ly", "level", "positive"):
        d[f] = src[f]
    d["score"], d["probability"] = src["fout"], src["prob"]
    return d


def _golden_worker(rank, world, port, nimages, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from featuredetection_amd import parallel
    total = []
    # two gathers (as bench.py does every --gather-every steps), detector id = pyramid layer of the record
    for lo, hi in ((0, nimages // 2), (nimages // 2, nimages)):
        mine = [i for i in parallel.shard_indices(nimages, rank, world) if lo <= i < hi]
        recs = [parallel.pack_records(np.full(len(d), i), d["layer"], d) for i in mine for d in [_golden_image_detections(i, nimages)]]
        local = np.concatenate(recs) if recs else np.zeros((0, parallel.RECORD_FIELDS))
        allr, trunc = parallel.gather_records(local, cap=512)
        assert not trunc
        total.append(allr)
    if rank == 0:
        q.put(np.concatenate(total))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_real_detection_records_world2():
    """world 2 over gloo with records packed from real fd_detection arrays (golden fixture): every field survives the gather
    ({image, detector, cx, cy, w, h, score, prob}), ordered by (image, detector, extraction order)"""
    from featuredetection_amd import parallel
    nimages, world = 9, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_golden_worker, args=(r, world, port, nimages, q)) for r in range(world)]
    for p in procs:
        p.start()
    allr = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    dets = [_golden_image_detections(i, nimages) for i in range(nimages)]
    assert sum(len(d) for d in dets) == 305 == len(allr)
    ref = np.concatenate([parallel.pack_records(np.full(len(d), i), d["layer"], d) for i, d in enumerate(dets)])
    halves = [ref[ref[:, 0] < nimages // 2], ref[ref[:, 0] >= nimages // 2]]
    exp = []
    for h in halves:
        exp.append(h[np.lexsort((np.arange(len(h)), h[:, 1], h[:, 0]))])
    exp = np.concatenate(exp)
    assert np.array_equal(allr, exp)
    # field fidelity: scores are float32 values, probabilities float64, geometry integers
    assert np.array_equal(allr[:, 6].astype(np.float32).astype(np.float64), allr[:, 6])
    assert np.array_equal(np.sort(allr[:, 7]), np.sort(np.concatenate([d["probability"] for d in dets])))
    assert np.array_equal(allr[:, 2:6], np.rint(allr[:, 2:6]))


def test_gather_single_process_and_truncation():
    from featuredetection_amd import parallel
    d = _detect_image(3)
    local = parallel.pack_records(np.full(len(d), 3), np.zeros(len(d)), d)
    allr, trunc = parallel.gather_records(local, cap=64)
    assert np.array_equal(allr, local) and not trunc
    big = np.zeros((10, parallel.RECORD_FIELDS))
    allr, trunc = parallel.gather_records(big, cap=4)
    assert trunc and len(allr) == 4


def test_native_record_packing_equals_the_python_twin():
    """fd_pack_records / fd_dist_owner (the C ABI's multi-GPU entry points, csrc/dist.hip) against parallel.pack_records /
    shard_indices on the committed fd_detection fixture: identical bytes, identical ownership (no GPU needed for these two)."""
    from featuredetection_amd import capi, parallel
    nimages = 9
    for i in range(nimages):
        d = _golden_image_detections(i, nimages)
        a = capi.pack_records(i, 3, d)
        b = parallel.pack_records(np.full(len(d), i), np.full(len(d), 3), d)
        assert a.shape == b.shape and a.tobytes() == b.tobytes()
    for world in (1, 2, 4, 8):
        for r in range(world):
            assert parallel.shard_indices(37, r, world) == [i for i in range(37) if capi.lib().fd_dist_owner(i, world) == r]


def test_config5_image_partition():
    """bench.py's config-5 job: image j belongs to rank j mod N (fd_dist_owner), a step = the job's images [256 i, 256 (i + 1)); over
    the ranks every image is fed exactly once, in its step, whatever N"""
    import bench
    for total in (10000, 24, 257, 1):
        steps = (total + 255) // 256
        for world in (1, 2, 3, 8):
            seen = []
            for i in range(steps):
                per_rank = [bench.Config5.step_images(i, r, world, total) for r in range(world)]
                for r, ids in enumerate(per_rank):
                    assert all(j % world == r for j in ids)
                    assert all(256 * i <= j < min(256 * (i + 1), total) for j in ids)
                seen.extend(np.concatenate(per_rank).tolist())
            assert sorted(seen) == list(range(total))
