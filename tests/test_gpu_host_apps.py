"""The reference-shaped C++ host layer (featuredetection_amd/host) driven through its example apps:
ffp_detect_app mirrors ffpDetectApp's object graph (config 1 plumbing), sdm_fit_app the two SDM calls.
Their printed results must equal the oracle's."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "featuredetection_amd")


def _run(args):
    env = dict(os.environ, LD_LIBRARY_PATH=PKG + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run(args, capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    return r.stdout


FACE_CFG = """detectors
{
    FaceFrontal
    {
        landmark "face"
        type fiveStageCascade ; same keys as ffpDetectApp/FaceFrontal.cfg
        firstClassifier pwvm
        {
            classifierFile %s
        }
        secondClassifier psvm
        {
            classifierFile %s
        }
        pyramid
        {
            minScaleFactor 0.05
            maxScaleFactor 0.16
            incrementalScaleFactor 0.92
            patch
            {
                width 20
                height 20
            }
        }
        overlapElimination
        {
            dist 5.0
            ratio 0.0
        }
    }
}
"""


def _read_patches(path):
    """the file ffp_detect_app --patches writes: per printed detection rows, cols, type (int32) + the pixels of getPatch()->getData()"""
    raw = open(path, "rb").read()
    out, k = [], 0
    while k < len(raw):
        rows, cols, typ = np.frombuffer(raw, np.int32, 3, k)
        k += 12
        dt = {0: np.uint8, 5: np.float32}[int(typ) & 7]   # CV_8U / CV_32F, one channel
        n = int(rows) * int(cols) * np.dtype(dt).itemsize
        out.append(np.frombuffer(raw, dt, int(rows) * int(cols), k).reshape(int(rows), int(cols)).copy())
        k += n
    return out


def test_ffp_detect_app_matches_oracle(tmp_path, oracle, synth, frame640, small_models):
    app = os.path.join(PKG, "ffp_detect_app")
    if not os.path.exists(app):
        pytest.fail("host apps not built (make -C featuredetection_amd/host)")
    wvm, svm = small_models
    synth.save_wvm(str(tmp_path / "face.fdwvm"), wvm)
    synth.save_svm_text(str(tmp_path / "face.svm.txt"), svm, rows=20, cols=20)
    synth.save_pnm(str(tmp_path / "frame.ppm"), frame640)
    cfg = FACE_CFG % (tmp_path / "face.fdwvm", tmp_path / "face.svm.txt")
    (tmp_path / "face.cfg").write_text(cfg)
    out = _run([app, str(tmp_path / "face.cfg"), str(tmp_path / "frame.ppm")])
    got = [l.split() for l in out.strip().splitlines()]
    po = oracle.Pyramid(inc=float(np.float32(0.92)), min_scale=float(np.float32(0.05)), max_scale=float(np.float32(0.16)))
    po.update(frame640)
    dets, stages = oracle.five_stage(po, oracle.Wvm(wvm), oracle.Svm(svm), 5.0, 0.0, 1, 1, None)
    assert len(got) == len(dets) > 0
    for g, d in zip(got, dets):
        assert g[0] == "FaceFrontal" and g[1] == "face"
        # Patch::getBounds (Patch.hpp:28-35)
        assert [int(v) for v in g[2:6]] == [d["cx"] - d["w"] // 2, d["cy"] - d["h"] // 2, d["w"], d["h"]]
        assert float(g[6]) == d["prob"]
    # Detector::keepPatchData(true) (--patches): the same detections, and every returned patch carries the pixels the reference's patch
    # carries -- the HistEq64-filtered 20x20 cut of its pyramid layer (Patch.hpp:28-243, DirectPyramidFeatureExtractor.cpp:75-123)
    out2 = _run([app, "--patches", str(tmp_path / "patches.bin"), str(tmp_path / "face.cfg"), str(tmp_path / "frame.ppm")])
    assert out2 == out
    patches = _read_patches(str(tmp_path / "patches.bin"))
    assert len(patches) == len(dets)
    for pt, d in zip(patches, dets):
        lp, lx, ly = oracle.extract_single(po, 20, 20, d["cx"], d["cy"], d["w"], d["h"])[:3]
        assert pt.dtype == np.uint8 and np.array_equal(pt, oracle.histeq64(np.ascontiguousarray(po.layer(lp)[ly:ly + 20, lx:lx + 20])))



def test_ffp_detect_app_image_sequence_detect_frames(tmp_path, oracle, synth, frame640, small_models):
    """FiveStageSlidingWindowDetector::detectFrames (backend extension of the C++ mirror: all images of a sequence through one
    multi-frame pyramid, one cascade run, one SVM launch): every frame's boxes and probabilities equal the oracle's detect(image)."""
    app = os.path.join(PKG, "ffp_detect_app")
    if not os.path.exists(app):
        pytest.fail("host apps not built (make -C featuredetection_amd/host)")
    wvm, svm = small_models
    synth.save_wvm(str(tmp_path / "face.fdwvm"), wvm)
    synth.save_svm_text(str(tmp_path / "face.svm.txt"), svm, rows=20, cols=20)
    frames = [frame640, synth.make_frame(640, 480, seed=31), synth.make_frame(640, 480, seed=32)]
    for i, f in enumerate(frames):
        synth.save_pnm(str(tmp_path / ("frame%d.ppm" % i)), f)
    cfg = FACE_CFG % (tmp_path / "face.fdwvm", tmp_path / "face.svm.txt")
    (tmp_path / "face.cfg").write_text(cfg)
    out = _run([app, str(tmp_path / "face.cfg")] + [str(tmp_path / ("frame%d.ppm" % i)) for i in range(len(frames))])
    got = [l.split() for l in out.strip().splitlines()]
    po = oracle.Pyramid(inc=float(np.float32(0.92)), min_scale=float(np.float32(0.05)), max_scale=float(np.float32(0.16)))
    wo, so = oracle.Wvm(wvm), oracle.Svm(svm)
    total = 0
    for fi, f in enumerate(frames):
        po.update(f)
        dets, _ = oracle.five_stage(po, wo, so, 5.0, 0.0, 1, 1, None)
        mine = [g for g in got if g[0] == "frame" and int(g[1]) == fi]
        assert len(mine) == len(dets)
        total += len(dets)
        for g, d in zip(mine, dets):
            assert g[2] == "FaceFrontal" and g[3] == "face"
            assert [int(v) for v in g[4:8]] == [d["cx"] - d["w"] // 2, d["cy"] - d["h"] // 2, d["w"], d["h"]]
            assert float(g[8]) == d["prob"]
    assert total > 0 and len(got) == total
    # --gpus 1: the image-shard path (one process per GPU, records through fd_dist_gather_records) prints the same lines
    out1 = _run([app, "--gpus", "1", str(tmp_path / "face.cfg")] + [str(tmp_path / ("frame%d.ppm" % i)) for i in range(len(frames))])
    assert out1 == out


def test_native_gather_world1_equals_python_twin(capi, ctx):
    """fd_dist_* on one GPU without a communicator (id == NULL): a world of one gathers from itself and orders the records like
    parallel.gather_records (image, detector, original order); truncation is reported; a gathered set that is only counted stays in the
    handle until it is fetched or dropped (fd_dist_gather_discard)"""
    from featuredetection_amd import parallel
    d = capi.Dist(ctx, 0, 1, None)
    rng = np.random.default_rng(5)
    local = np.zeros((300, 8))
    local[:, 0] = rng.integers(0, 12, 300)   # image ids out of order
    local[:, 1] = rng.integers(0, 3, 300)
    local[:, 2:] = rng.normal(size=(300, 6))
    got, tr = d.gather(local, 512)
    ref, tr2 = parallel.gather_records(local, 512)
    assert not tr and not tr2 and got.tobytes() == ref.tobytes()
    got, tr = d.gather(local, 100)
    ref, tr2 = parallel.gather_records(local, 100)
    assert tr and tr2 and got.tobytes() == ref.tobytes()
    got, tr = d.gather(np.zeros((0, 8)), 16)
    assert len(got) == 0 and not tr
    # count-only call: the set waits in the handle; a follow-up call delivers IT (its own local records are ignored) ...
    assert capi.dist_gather_count(d, local, 512) == 300 and d.pending()
    got, _ = d.gather(local[:7], 512)
    assert len(got) == 300 and not d.pending()
    # ... unless the caller drops it: the next call is a new collective over the new records
    assert capi.dist_gather_count(d, local, 512) == 300 and d.pending()
    d.discard()
    assert not d.pending()
    got, _ = d.gather(local[:7], 512)
    assert len(got) == 7
    d.close()


REAL_RCCL_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import torch  # noqa: F401  (its HIP runtime first, like every GPU test)
from featuredetection_amd import capi, parallel
ctx = capi.Context(0)
uid = capi.Dist.unique_id()                 # ncclGetUniqueId of the box's librccl.so
assert len(uid) == 128 and any(uid)
real = capi.Dist(ctx, 0, 1, uid)            # ncclCommInitRank(comm, 1, id, 0): a real one-rank communicator
plain = capi.Dist(ctx, 0, 1, None)          # no communicator: the rank gathers from itself
rng = np.random.default_rng(5)
for n, cap in ((300, 512), (300, 100), (0, 16), (4096, 4096)):
    local = np.zeros((n, 8))
    local[:, 0] = rng.integers(0, 12, n); local[:, 1] = rng.integers(0, 3, n); local[:, 2:] = rng.normal(size=(n, 6))
    a, ta = real.gather(local, cap)         # hipMemcpyAsync -> ncclAllGather on the context's stream -> hipMemcpyAsync -> sync
    b, tb = plain.gather(local, cap)
    c, tc = parallel.gather_records(local, cap)
    assert ta == tb == tc and a.tobytes() == b.tobytes() == c.tobytes(), (n, cap)
# two collectives back to back on the same communicator, then the count-then-fetch form (ONE collective)
assert capi.dist_gather_count(real, local, 4096) == 4096 and real.pending()
a, _ = real.gather(local[:3], 4096)
assert len(a) == 4096
# the two-halves form: the payload travels on the handle's own stream between _begin and _end
real.gather_begin(local[:77], 4096)
plain.gather_begin(local[:77], 4096)
a, ta = real.gather_end()
b, tb = plain.gather_end()
assert len(a) == 77 and a.tobytes() == b.tobytes() and not ta and not tb
real.gather_begin(np.zeros((0, 8)), 16)     # nobody has a record: the payload collective is skipped
a, _ = real.gather_end()
assert len(a) == 0
real.close(); plain.close()
print("REAL_RCCL_OK")
"""


def test_real_rccl_one_rank_communicator():
    """VERDICT r04 task 7: the product's librccl binding against the REAL library on the test box -- ncclGetUniqueId ->
    ncclCommInitRank(world = 1) -> ncclAllGather on the context's stream (csrc/dist.hip: the branch N > 1 ranks take), records equal
    the no-communicator path and the torch twin.  Proves symbol binding, sizeof(ncclUniqueId), the call signatures and the stream
    semantics; more than one physical GPU stays unmeasured here.  In a child process with a timeout: a communicator that cannot
    bootstrap must fail this test, not hang the suite."""
    env = dict(os.environ, FD_DIST_FORCE_COMM="1")   # a one-rank communicator is an explicit request
    env.pop("FD_RCCL_LIB", None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        r = subprocess.run([sys.executable, "-c", REAL_RCCL_SCRIPT % ROOT], env=env, capture_output=True, text=True, timeout=240)
    except subprocess.TimeoutExpired as e:
        pytest.fail("real librccl one-rank communicator did not finish in 240 s: %s" % ((e.stderr or b"")[-2000:],))
    assert r.returncode == 0 and "REAL_RCCL_OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])


STUB_RCCL = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stub_rccl", "librccl_stub.so")


def test_ffp_detect_app_two_ranks_on_one_gpu(tmp_path, synth, frame640, small_models):
    """ffp_detect_app --gpus 2 as two processes on GPU 0 (FD_DIST_ONE_DEVICE=1) with the test-only librccl stand-in (tests/stub_rccl:
    the ranks meet in shared memory): the multi-rank branch of fd_dist_init / fd_dist_gather_records, the unique-id hand-off through a
    file and the rank launcher run for real; the lines rank 0 prints are the single process's.  A rank that cannot do its part (an
    unreadable image) ends the run with an error instead of leaving the other rank in the collective."""
    app = os.path.join(PKG, "ffp_detect_app")
    if not os.path.exists(app):
        pytest.fail("host apps not built (make -C featuredetection_amd/host)")
    if not os.path.exists(STUB_RCCL):
        pytest.fail("tests/stub_rccl/librccl_stub.so not built (__graft_entry__.build() / make -C tests/stub_rccl)")
    wvm, svm = small_models
    synth.save_wvm(str(tmp_path / "face.fdwvm"), wvm)
    synth.save_svm_text(str(tmp_path / "face.svm.txt"), svm, rows=20, cols=20)
    frames = [frame640] + [synth.make_frame(640, 480, seed=40 + i) for i in range(4)]
    names = []
    for i, f in enumerate(frames):
        names.append(str(tmp_path / ("frame%d.ppm" % i)))
        synth.save_pnm(names[-1], f)
    (tmp_path / "face.cfg").write_text(FACE_CFG % (tmp_path / "face.fdwvm", tmp_path / "face.svm.txt"))
    single = _run([app, str(tmp_path / "face.cfg")] + names)
    assert len(single.strip().splitlines()) > 0
    env = dict(os.environ, LD_LIBRARY_PATH=PKG + ":" + os.environ.get("LD_LIBRARY_PATH", ""), FD_RCCL_LIB=STUB_RCCL, FD_DIST_ONE_DEVICE="1",
               FD_DIST_TIMEOUT_S="120")
    import subprocess
    r = subprocess.run([app, "--gpus", "2", str(tmp_path / "face.cfg")] + names, capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    assert r.stdout == single
    r3 = subprocess.run([app, "--gpus", "3", str(tmp_path / "face.cfg")] + names, capture_output=True, text=True, env=env, timeout=300)
    assert r3.returncode == 0 and r3.stdout == single, r3.stderr
    # failure modes: reported before any rank starts, nothing hangs
    bad = subprocess.run([app, "--gpus", "2", str(tmp_path / "face.cfg")] + names[:2] + [str(tmp_path / "missing.ppm")],
                         capture_output=True, text=True, env=env, timeout=120)
    assert bad.returncode != 0 and "cannot open image" in bad.stderr
    env2 = dict(env)
    env2.pop("FD_DIST_ONE_DEVICE")
    many = subprocess.run([app, "--gpus", "64", str(tmp_path / "face.cfg")] + names, capture_output=True, text=True, env=env2, timeout=120)
    assert many.returncode != 0 and "device(s) are visible" in many.stderr


def _dist_rank_main(rank, world, uid_hex, cap, q):
    """a rank of test_native_gather_two_ranks (own process: own HIP runtime and context)"""
    import numpy as np
    import torch  # noqa: F401  (before libfd_hip.so)
    from featuredetection_amd import capi as C
    ctx = C.Context(0)
    d = C.Dist(ctx, rank, world, bytes.fromhex(uid_hex))
    rng = np.random.default_rng(100 + rank)
    local = np.zeros((50 + 30 * rank, 8))
    local[:, 0] = rng.integers(0, 9, len(local)) * world + rank
    local[:, 1] = rng.integers(0, 3, len(local))
    local[:, 2:] = rng.normal(size=(len(local), 6))
    got, tr = d.gather(local, cap)
    got2, tr2 = d.gather(local[:7], cap)   # a second collective on the same communicator
    # count-then-fetch: the collective runs in the first call, the second one only delivers
    n = C.dist_gather_count(d, local, cap)
    got3, _ = d.gather(np.zeros((0, 8)), cap)
    q.put((rank, local, got, tr, got2, n, got3))
    d.close()


def test_native_gather_two_ranks(capi, ctx):
    """capi.Dist with world 2: two processes on GPU 0 and the librccl stand-in.  Every rank receives the records of both, ordered like
    the torch.distributed twin orders them; a count-only call followed by a fetch is ONE collective."""
    if not os.path.exists(STUB_RCCL):
        pytest.fail("tests/stub_rccl/librccl_stub.so not built")
    import multiprocessing as mp
    from featuredetection_amd import parallel
    os.environ["FD_RCCL_LIB"] = STUB_RCCL
    try:
        mpc = mp.get_context("spawn")
        uid = capi.Dist.unique_id()   # loads the stand-in in this process as well
        q = mpc.Queue()
        ps = [mpc.Process(target=_dist_rank_main, args=(r, 2, uid.hex(), 256, q)) for r in range(2)]
        for p_ in ps:
            p_.start()
        res = sorted([q.get(timeout=240) for _ in ps], key=lambda t: t[0])
        for p_ in ps:
            p_.join(timeout=60)
            assert p_.exitcode == 0
    finally:
        os.environ.pop("FD_RCCL_LIB", None)
    both = np.concatenate([res[0][1], res[1][1]])
    order = np.lexsort((np.arange(len(both)), both[:, 1], both[:, 0]))   # (image, detector, original order)
    exp = both[order]
    for rank, local, got, tr, got2, n, got3 in res:
        assert not tr and got.tobytes() == exp.tobytes(), rank
        b2 = np.concatenate([res[0][1][:7], res[1][1][:7]])
        assert got2.tobytes() == b2[np.lexsort((np.arange(len(b2)), b2[:, 1], b2[:, 0]))].tobytes()
        assert n == len(exp) and got3.tobytes() == exp.tobytes()


SINGLE_CFG = """detectors
{
    Face
    {
        landmark "face"
        type single
        feature %s
        classifier psvm
        {
            classifierFile %s
            threshold %s
        }
        pyramid
        {
            minScaleFactor 0.2
            maxScaleFactor 0.4
            incrementalScaleFactor 0.7071
            patch
            {
                width 20
                height 20
            }
        }
    }
}
"""


@pytest.mark.parametrize("feature", ["whi", "histeq", "gray"])
def test_ffp_detect_app_single_detector_feature_spaces(tmp_path, oracle, synth, frame640, feature):
    """type "single" of ffpDetectApp.cpp:427-500: feature spaces whi / histeq / gray with a ProbabilisticSvmClassifier,
    through the C++ mirror classes (WhiteningFilter ... UnitNormFilter, HistogramEqualizationFilter)."""
    app = os.path.join(PKG, "ffp_detect_app")
    if not os.path.exists(app):
        pytest.fail("host apps not built (make -C featuredetection_amd/host)")
    small = np.ascontiguousarray(frame640[:240, :320])
    inc = float(np.float32(0.7071))
    po = oracle.Pyramid(inc=inc, min_scale=float(np.float32(0.2)), max_scale=float(np.float32(0.4)))
    po.update(small)
    layers = [po.layer(i) for i in range(len(po.layers()))]
    wins = po.windows(20, 20, 1, 1)
    pat = np.stack([np.ascontiguousarray(layers[lp][ly:ly + 20, lx:lx + 20]) for lp, lx, ly, *_ in wins])
    rng = np.random.default_rng(8)
    if feature == "whi":
        feats = np.stack([oracle.whi(p_).ravel() for p_ in pat])
        sv = feats[rng.choice(len(feats), 48, replace=False)].copy()
        m = dict(kernel=2, dtype=1, sv=sv, p0=2.0)
    else:
        feats = (np.stack([oracle.equalize_hist(p_) for p_ in pat]) if feature == "histeq" else pat).reshape(len(pat), -1)
        sv = feats[rng.choice(len(feats), 48, replace=False)].copy()
        m = dict(kernel=2, dtype=0, sv=sv, p0=2e-6)
    m.update(coeff=rng.normal(0, 1, 48).astype(np.float32), bias=np.float32(0.05), p1=0.0, p2=0.0, threshold=0.0, logistic_a=0.3, logistic_b=-1.7)
    so = oracle.Svm(m)
    do = so.distance(feats)
    order = np.sort(do)
    # threshold in the widest gap of the upper tail, so that no distance sits within rounding of it
    tail = order[-60:]
    k = int(np.argmax(np.diff(tail)))
    m["threshold"] = float(np.float32(0.5 * (tail[k] + tail[k + 1])))
    so = oracle.Svm(m)
    synth.save_svm_text(str(tmp_path / "c.svm.txt"), m, rows=20, cols=20)
    synth.save_pnm(str(tmp_path / "frame.ppm"), small)
    (tmp_path / "c.cfg").write_text(SINGLE_CFG % (feature, tmp_path / "c.svm.txt", repr(m["threshold"])))
    out = _run([app, str(tmp_path / "c.cfg"), str(tmp_path / "frame.ppm")])
    got = [l.split() for l in out.strip().splitlines()]
    pos = np.nonzero(do >= m["threshold"])[0]
    assert len(got) == len(pos) > 0
    for g, i in zip(got, pos):
        lp, lx, ly, cx, cy, ow, oh = [int(v) for v in wins[i]]
        assert [int(v) for v in g[2:6]] == [cx - ow // 2, cy - oh // 2, ow, oh]
        assert abs(float(g[6]) - so.probability(do[i])) <= 1e-6
    # --patches: the patch of every detection after the feature space's filter chain, composed per Mat
    out2 = _run([app, "--patches", str(tmp_path / "patches.bin"), str(tmp_path / "c.cfg"), str(tmp_path / "frame.ppm")])
    assert out2 == out
    patches = _read_patches(str(tmp_path / "patches.bin"))
    assert len(patches) == len(pos)
    for pt, i in zip(patches, pos):
        if feature == "whi":
            assert pt.dtype == np.float32 and np.allclose(pt.ravel(), feats[i], rtol=1e-6, atol=1e-9)
        else:
            assert pt.dtype == np.uint8 and np.array_equal(pt.ravel(), feats[i])

def test_sdm_fit_app_matches_oracle(tmp_path, oracle, synth):
    app = os.path.join(PKG, "sdm_fit_app")
    if not os.path.exists(app):
        pytest.fail("host apps not built (make -C featuredetection_amd/host)")
    model = synth.make_sdm(4, L=20, S=3)
    # the text file stores 9 significant digits: reload what the app will read
    synth.save_sdm_text(str(tmp_path / "sdm.txt"), model)
    gray = synth.make_frame(200, 180, seed=55, channels=1)
    synth.save_pnm(str(tmp_path / "face.pgm"), gray)
    out = _run([app, str(tmp_path / "sdm.txt"), str(tmp_path / "face.pgm"), "30", "25", "120", "130", str(tmp_path / "lms.txt")])
    got = np.array([float(v) for v in out.split()], np.float32)
    st, ref = oracle.sdm_fit(gray, model, [30, 25, 120, 130])
    assert st == 0 and got.shape == ref.shape
    assert np.allclose(got, ref, rtol=1e-4, atol=1e-4), np.abs(got - ref).max()
    # imageio::SimpleModelLandmarkSink: "name x y" per landmark (SimpleModelLandmarkSink.cpp:29-31)
    lines = [l.split() for l in (tmp_path / "lms.txt").read_text().strip().splitlines()]
    assert [l[0] for l in lines] == [str(i) for i in range(20)]
    xy = np.array([[float(l[1]), float(l[2])] for l in lines], np.float32)
    assert np.allclose(xy[:, 0], got[:20], rtol=1e-5) and np.allclose(xy[:, 1], got[20:], rtol=1e-5)


def test_sdm_fit_app_real_model_non_adaptive(tmp_path, oracle, synth):
    """The reference's trained model (tests/golden/sdm_real_11012014.npz) written in the CURRENT text format with its
    descriptor parameters (`descriptorParameters numCells n cellSize c numBins b`, SdmLandmarkModel.cpp:188-204), loaded by the host
    layer's SdmLandmarkModel::load and fitted by SdmLandmarkModelFitting(model, adaptive = false): the landmarks of sdm_fit_app
    --non-adaptive equal the oracle's, and without the switch the 144-value regressors do not fit the adaptive 279-value descriptors (an
    error, like the reference's gemm assertion)."""
    import importlib.util
    app = os.path.join(PKG, "sdm_fit_app")
    if not os.path.exists(app):
        pytest.fail("host apps not built (make -C featuredetection_amd/host)")
    G = os.path.join(ROOT, "tests", "golden")
    spec = importlib.util.spec_from_file_location("make_sdm_real", os.path.join(G, "make_sdm_real.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    g = np.load(os.path.join(G, "sdm_real_11012014.npz"))
    model = mod.unpack_model(g)
    synth.save_sdm_text(str(tmp_path / "real.txt"), model)
    f = 5
    gray = synth.make_frame(256, 256, seed=int(g["frame_seed0"]) + f, channels=1)
    synth.save_pnm(str(tmp_path / "face.pgm"), gray)
    box = [str(int(v)) for v in g["face_box"]]
    out = _run([app, "--non-adaptive", str(tmp_path / "real.txt"), str(tmp_path / "face.pgm")] + box)
    got = np.array([float(v) for v in out.split()], np.float32)
    ref = g["oracle_shapes"][f, model["S"]]
    assert got.shape == ref.shape and np.allclose(got, ref, rtol=1e-4, atol=1e-4), np.abs(got - ref).max()
    env = dict(os.environ, LD_LIBRARY_PATH=PKG + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([app, str(tmp_path / "real.txt"), str(tmp_path / "face.pgm")] + box, capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "regressor" in (r.stderr + r.stdout)


def test_ffp_detect_app_prvm_single_detector(tmp_path, oracle, synth, frame640):
    """type "single" with classifier "prvm" (ffpDetectApp.cpp:427-500): hq64 feature space + conversionFilter patch
    filter + ProbabilisticRvmClassifier, through the C++ mirror classes and fd_detect_rvm."""
    app = os.path.join(PKG, "ffp_detect_app")
    if not os.path.exists(app):
        pytest.fail("host apps not built (make -C featuredetection_amd/host)")
    small = np.ascontiguousarray(frame640[:240, :320])
    po = oracle.Pyramid(inc=float(np.float32(0.7071)), min_scale=float(np.float32(0.2)), max_scale=float(np.float32(0.4)))
    po.update(small)
    layers = [po.layer(i) for i in range(len(po.layers()))]
    wins = po.windows(20, 20, 1, 1)
    pat = np.stack([oracle.histeq64(np.ascontiguousarray(layers[lp][ly:ly + 20, lx:lx + 20])) for lp, lx, ly, *_ in wins])
    scale = 0.25
    feats = pat.reshape(len(pat), -1).astype(np.float32) * np.float32(scale)
    m = synth.make_rvm(12, feats[::4], 20, 20, n_filters=20, kernel=2)
    m["logistic_a"], m["logistic_b"] = 0.4, -2.0
    synth.save_rvm(str(tmp_path / "c.fdrvm"), m)
    synth.save_pnm(str(tmp_path / "frame.ppm"), small)
    cfg = """detectors
{
    Face
    {
        landmark "face"
        type single
        feature hq64
        patchFilter
        {
            conversionFilter "5 0.25"
        }
        classifier prvm
        {
            classifierFile %s
            logisticA 0.4
            logisticB -2.0
        }
        pyramid
        {
            minScaleFactor 0.2
            maxScaleFactor 0.4
            incrementalScaleFactor 0.7071
            patch
            {
                width 20
                height 20
            }
        }
    }
}
""" % (tmp_path / "c.fdrvm")
    (tmp_path / "c.cfg").write_text(cfg)
    out = _run([app, str(tmp_path / "c.cfg"), str(tmp_path / "frame.ppm")])
    got = [l.split() for l in out.strip().splitlines()]
    ro = oracle.Rvm(m)
    lv, dd = ro.eval(feats)
    pos = np.nonzero((lv == 19) & (dd >= m["thresholds"][19]))[0]
    assert len(got) == len(pos) > 0
    for g, i in zip(got, pos):
        lp, lx, ly, cx, cy, ow, oh = [int(v) for v in wins[i]]
        assert [int(v) for v in g[2:6]] == [cx - ow // 2, cy - oh // 2, ow, oh]
        assert abs(float(g[6]) - ro.probability(dd[i])) <= 1e-9


def test_condensation_eval_app_matches_oracle(tmp_path, oracle, synth, frame640, small_models):
    """condensation::WvmSvmModel (C++ mirror) through condensation_eval_app vs the oracle's restatement of WvmSvmModel.cpp:69-118."""
    app = os.path.join(PKG, "condensation_eval_app")
    if not os.path.exists(app):
        pytest.fail("host apps not built (make -C featuredetection_amd/host)")
    wvm, svm = small_models
    synth.save_wvm(str(tmp_path / "face.fdwvm"), wvm)
    synth.save_svm_text(str(tmp_path / "face.svm.txt"), svm, rows=20, cols=20)
    synth.save_pnm(str(tmp_path / "frame.ppm"), frame640)
    cfg = """detectors
{
    FaceFrontal
    {
        firstClassifier pwvm
        {
            classifierFile %s
        }
        secondClassifier psvm
        {
            classifierFile %s
            threshold %s
        }
        pyramid
        {
            minScaleFactor 0.05
            maxScaleFactor 0.16
            incrementalScaleFactor 0.92
            patch
            {
                width 20
                height 20
            }
        }
    }
}
""" % (tmp_path / "face.fdwvm", tmp_path / "face.svm.txt", repr(float(svm["threshold"])))
    (tmp_path / "face.cfg").write_text(cfg)
    po = oracle.Pyramid(inc=float(np.float32(0.92)), min_scale=float(np.float32(0.05)), max_scale=float(np.float32(0.16)))
    po.update(frame640)
    wo, so = oracle.Wvm(wvm), oracle.Svm(svm)
    pos, _, _ = oracle.sliding_wvm(po, wo, 1, 1)
    rng = np.random.default_rng(3)
    size = rng.integers(100, 440, 400)
    samples = np.stack([rng.integers(0, 640, 400), rng.integers(0, 480, 400), size], 1)
    samples = np.concatenate([samples, np.array([[d["cx"], d["cy"], d["w"]] for d in pos[:20]])]).astype(np.int32)
    (tmp_path / "samples.txt").write_text("".join("%d %d %d\n" % tuple(r) for r in samples))
    out = _run([app, str(tmp_path / "face.cfg"), str(tmp_path / "frame.ppm"), str(tmp_path / "samples.txt")])
    got = np.array([[float(v) for v in l.split()] for l in out.strip().splitlines()])
    to, wt = oracle.wvm_svm_evaluate(po, wo, so, np.concatenate([samples, samples[:, 2:3]], 1))   # Sample::aspectRatio = 1
    assert len(got) == len(samples)
    assert np.array_equal(got[:, 0].astype(bool), to) and (wt > 0).sum() > 20
    assert np.allclose(got[:, 1], wt, rtol=1e-12, atol=0)


def test_host_selftest_app(tmp_path, oracle, capi, ctx, synth, frame640):
    """Pyramid-on-pyramid views, layer sub-ranges + ROI on a fused chain, FilteringPyramidFeatureExtractor and the stand-alone
    applyTo of every filter, through the reference-shaped C++ classes (host_selftest_app) against the oracle."""
    app = os.path.join(PKG, "host_selftest_app")
    if not os.path.exists(app):
        pytest.fail("host apps not built (make -C featuredetection_amd/host)")
    gray = oracle.bgr2gray(frame640)[:300, :400].copy()
    synth.save_pnm(str(tmp_path / "g.pgm"), gray)
    out = _run([app, str(tmp_path / "g.pgm"), str(tmp_path)])
    kv = {l.split()[0] + ("_" + l.split()[1] if l.split()[0] == "view" else ""): l.split() for l in out.strip().splitlines()}
    lines = out.strip().splitlines()
    # 1. fifteen extractors on views of one source pyramid: built once per image version
    assert "builds 1" in lines and "builds_after_new_version 2" in lines, out
    po = oracle.Pyramid(inc=0.9, min_scale=0.09, max_scale=0.7)
    po.update(gray)
    assert "source_layers %d" % len(po.layers()) in lines
    for i, (lo, hi, pw) in enumerate(((0.09, 0.25, 20), (0.5, 0.7, 24), (0.3, 0.45, 20))):
        sel = [k for k, l in enumerate(po.layers()) if lo <= l["scale"] <= hi]
        assert "view %d layers %d first %d last %d" % (i, len(sel), po.layers()[sel[0]]["index"], po.layers()[sel[-1]]["index"]) in lines
        wins = po.windows(pw, pw, 4, 4)
        wins = wins[np.isin(wins[:, 0], sel)]
        geo = np.fromfile(str(tmp_path / ("view%d_geo.bin" % i)), np.int32).reshape(-1, 4)
        assert np.array_equal(geo, wins[:, 3:7]), i
        # stepLayer = 2 counts from the view's first layer (views 1 and 2 start in the middle of the source pyramid)
        geo2 = np.fromfile(str(tmp_path / ("view%d_step2_geo.bin" % i)), np.int32).reshape(-1, 4)
        assert len(sel[::2]) < len(sel) and np.array_equal(geo2, wins[np.isin(wins[:, 0], sel[::2])][:, 3:7]), i
    # 2. layer sub-range (first, last, step 2) + ROI on the HOG chain: geometry and features
    ph = oracle.Pyramid(octave_layers=3, min_scale=0.2, max_scale=0.8)
    ph.set_layer_filter(1, bins=9)
    ph.update(gray)
    L = ph.layers()
    first, last = L[1]["index"], L[-2]["index"]
    sel = [k for k, l in enumerate(L) if k % 2 == 0 and first <= l["index"] <= last]
    wins = ph.windows(20, 20, 3, 3, (40, 30, 200, 160))
    wins = wins[np.isin(wins[:, 0], sel)]
    assert len(wins) > 20
    geo = np.fromfile(str(tmp_path / "hog_sub_geo.bin"), np.int32).reshape(-1, 4)
    assert np.array_equal(geo, wins[:, 3:7])
    feat = np.fromfile(str(tmp_path / "hog_sub_feat.bin"), np.float32).reshape(len(wins), -1)
    layers = [ph.layer(k) for k in range(len(L))]
    exp = np.stack([oracle.hog_filter(np.ascontiguousarray(layers[w[0]][w[2]:w[2] + 20, w[1]:w[1] + 20]), 9, 5, 2) for w in wins])
    assert np.array_equal(feat, exp)
    # 3. FilteringPyramidFeatureExtractor: the whi chain is fused and equals the per-Mat composition; an unfused chain composes per Mat
    fl = [l for l in lines if l.startswith("filtering")][0].split()
    assert fl[2] == "1" and int(fl[4]) > 0 and fl[-1] == "0" and int(fl[6]) > 0, fl
    gl = [l for l in lines if l.startswith("generic")][0].split()
    assert gl[2] == "0" and int(gl[4]) > 0
    raw0 = np.fromfile(str(tmp_path / "raw_patch0.bin"), np.uint8).reshape(20, 20)
    assert np.array_equal(np.fromfile(str(tmp_path / "generic_lbp_patch0.bin"), np.uint8).reshape(20, 20), oracle.lbp(raw0, 0))
    # 4. stand-alone applyTo of every filter
    crop = np.fromfile(str(tmp_path / "crop.bin"), np.uint8).reshape(48, 64)
    assert np.array_equal(crop, gray[24:72, 16:80])
    grad = oracle.gradient_filter(crop, 3)
    assert np.array_equal(np.fromfile(str(tmp_path / "grad.bin"), np.uint8).reshape(48, 64, 2), grad)
    bins = oracle.gradient_binning(grad, 9, False, True)
    assert np.array_equal(np.fromfile(str(tmp_path / "bins.bin"), np.uint8).reshape(48, 64, 4), bins)
    assert np.array_equal(np.fromfile(str(tmp_path / "lbp.bin"), np.uint8).reshape(48, 64), oracle.lbp(crop, 1))
    p20 = np.ascontiguousarray(bins[4:24, 4:24])
    f32 = lambda name: np.fromfile(str(tmp_path / name), np.float32)
    assert np.array_equal(f32("hog.bin"), np.asarray(oracle.hog_filter(p20, 9, 5, 2, interpolate=True), np.float32).ravel())
    assert np.array_equal(f32("sphist.bin"), np.asarray(oracle.spatial_histogram(p20, 9, 5, 2, interpolate=True, normalization=2), np.float32).ravel())
    assert np.array_equal(f32("phog.bin"), np.asarray(oracle.pyramid_hog(p20, 9, 2, interpolate=True), np.float32).ravel())
    assert np.array_equal(f32("sppyr.bin"), np.asarray(oracle.spatial_pyramid_histogram(p20, 9, 2, interpolate=True, normalization=3), np.float32).ravel())
    g20 = np.ascontiguousarray(crop[10:30, 10:30])
    assert np.array_equal(np.fromfile(str(tmp_path / "whitened.bin"), np.uint8).reshape(20, 20), oracle.whitening(g20))
    assert np.array_equal(f32("converted.bin"), (g20.astype(np.float32) * np.float32(1.0 / 255.0) + np.float32(0.0)).ravel())
    v = g20.astype(np.float32).ravel()
    inv = np.float32(1.0 / (np.sqrt((v.astype(np.float64) ** 2).sum()) + np.float64(np.float32(1e-4))))
    assert np.array_equal(f32("unitnorm.bin"), v * inv)
    assert "reshaped 1 x 400" in lines
    # ZeroMeanUnitVarianceFilter.cpp:21-34 (mean / population deviation in double) and FilteringFeatureExtractor applying it to one patch
    def zmuv(a):
        a = a.astype(np.float32).astype(np.float64)
        return ((a - a.mean()) / a.std()).astype(np.float32).ravel()
    assert np.array_equal(f32("zmuv.bin"), zmuv(g20))
    raw = np.fromfile(str(tmp_path / "ffe_raw.bin"), np.uint8)
    assert len(raw) == 400 and np.array_equal(f32("ffe_patch.bin"), zmuv(raw))


def test_svm_text_format_fixture(tmp_path, capi, ctx):
    """SvmClassifier::loadFromText (SvmClassifier.cpp:161-239) on a fixture written by hand in the reference's format (out-of-order
    alphas, FullPolynomial degree 2): the app's single psvm detector loads it through ProbabilisticSvmClassifier::load(ptree); its
    decision values are checked through the C ABI against the closed form (scale * <x, s> + constant)^degree."""
    here = os.path.dirname(os.path.abspath(__file__))
    txt = open(os.path.join(here, "golden", "svm_fullpolynomial.txt")).read().splitlines()
    assert txt[0].split() == ["FullPolynomial", "2", "1.5", "0.25"]
    sv = np.array([[float(v) for v in l.split()] for l in txt[8:11]], np.float32)
    coeff = np.zeros(3, np.float32)
    for l in txt[4:7]:
        i, a = l[len("alphas["):].split("]=")
        coeff[int(i)] = np.float32(a)
    m = dict(kernel=1, p0=0.25, p1=1.5, p2=2.0, dtype=1, sv=sv, coeff=coeff, bias=np.float32(0.125), threshold=0.0, logistic_a=0.00556, logistic_b=-2.95)
    x = np.array([[0.5, -1.0, 2.0, 0.25], [3.0, 1.0, 0.0, -2.0]], np.float32)
    d = capi.Svm(ctx, m).distance(x)
    exp = ((0.25 * (x.astype(np.float64) @ sv.astype(np.float64).T) + 1.5) ** 2) @ coeff.astype(np.float64) - 0.125
    assert np.allclose(d, exp, rtol=1e-6)
    # through the host layer: a "single" psvm detector on gray 2x2 patches reads the same file with the reference's loader
    app = os.path.join(PKG, "ffp_detect_app")
    img = np.zeros((40, 40), np.uint8)
    img[10:30, 10:30] = 200
    from featuredetection_amd import synth
    synth.save_pnm(str(tmp_path / "i.pgm"), img)
    cfg = """detectors
{
    Blob
    {
        landmark "face"
        type single
        feature gray
        patchFilter
        {
            reshapingFilter 1
            conversionFilter "5 0.00392156862745098"
        }
        classifier psvm
        {
            classifierFile %s
            threshold 0.0
        }
        pyramid
        {
            minScaleFactor 0.5
            maxScaleFactor 1.0
            incrementalScaleFactor 0.5
            patch
            {
                width 2
                height 2
            }
        }
    }
}
""" % os.path.join(here, "golden", "svm_fullpolynomial.txt")
    (tmp_path / "c.cfg").write_text(cfg)
    out = _run([app, str(tmp_path / "c.cfg"), str(tmp_path / "i.pgm")])
    got = sorted(tuple(int(v) for v in l.split()[2:6]) for l in out.strip().splitlines())
    # expected: every 2x2 window (step 1) of the oracle's pyramid, gray / 255 features, closed-form decision value >= 0
    from oracle import pyoracle as O
    po = O.Pyramid(inc=0.5, min_scale=0.5, max_scale=1.0)
    po.update(img)
    layers = [po.layer(k) for k in range(len(po.layers()))]
    exp_boxes = []
    for w in po.windows(2, 2, 1, 1):
        f = layers[w[0]][w[2]:w[2] + 2, w[1]:w[1] + 2].astype(np.float32).ravel() * np.float32(0.00392156862745098)
        dv = ((0.25 * (f.astype(np.float64) @ sv.astype(np.float64).T) + 1.5) ** 2) @ coeff.astype(np.float64) - 0.125
        if dv >= 0.0:
            exp_boxes.append((int(w[3] - w[5] // 2), int(w[4] - w[6] // 2), int(w[5]), int(w[6])))
    assert len(exp_boxes) > 0 and got == sorted(exp_boxes)


def test_bench_two_ranks_gather_through_fd_dist(tmp_path):
    """bench.py --gpus 2 the way the driver launches it (torch.distributed.run, one process per rank), on a one-GPU box: both ranks on
    device 0, torch.distributed over gloo, fd_dist_* over the librccl stand-in.  The records every rank produced arrive on rank 0
    through fd_dist_gather_records (records_gathered == detections_delivered), nothing is truncated."""
    import json
    import subprocess
    import sys
    if not os.path.exists(STUB_RCCL):
        pytest.fail("tests/stub_rccl/librccl_stub.so not built")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FD_DIST_ONE_DEVICE="1", FD_BENCH_DIST_BACKEND="gloo", FD_RCCL_LIB=STUB_RCCL)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29571",
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--also", "none", "--no-cpu-baseline", "--no-probe",
           "--frames-per-step", "256"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak"
    assert rec["records_gathered"] == rec["detections_delivered"] > 0 and not rec["records_truncated"]


def test_bench_config5_one_and_two_ranks(tmp_path):
    """BASELINE config 5 as a bench workload (VERDICT r04 task 1b), scaled down to 24 images: `bench.py --workload config5` on one rank,
    and on two ranks the way the driver launches it (both on device 0, gloo + the librccl stand-in).  The job is fixed -- every image
    is generated from (seed, image index) -- so both runs deliver the same number of detections, rank 0 receives all of them through
    fd_dist_gather_records, and the record says "strong"."""
    import json
    import subprocess
    import sys
    if not os.path.exists(STUB_RCCL):
        pytest.fail("tests/stub_rccl/librccl_stub.so not built")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tail = [os.path.join(root, "bench.py"), "--workload", "config5", "--images", "24", "--also", "none", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-probe"]
    r1 = subprocess.run([sys.executable] + tail, capture_output=True, text=True, timeout=900, cwd=root)
    assert r1.returncode == 0, r1.stderr[-2000:]
    rec1 = json.loads(r1.stdout.strip().splitlines()[-1])
    assert rec1["n_gpus"] == 1 and rec1["scaling"] == "strong" and rec1["steps"] == 1 and rec1["config"]["images"] == 24
    assert rec1["detections_delivered"] > 0 and "config 5" in rec1["config"]["workload"]
    env = dict(os.environ, FD_DIST_ONE_DEVICE="1", FD_BENCH_DIST_BACKEND="gloo", FD_RCCL_LIB=STUB_RCCL)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29573"] + tail + ["--gpus", "2"]
    r2 = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=root)
    assert r2.returncode == 0, r2.stderr[-2000:]
    rec2 = json.loads(r2.stdout.strip().splitlines()[-1])
    assert rec2["n_gpus"] == 2 and rec2["scaling"] == "strong" and rec2["steps"] == 1
    assert rec2["detections_delivered"] == rec1["detections_delivered"]          # the same 24 images, whatever the sharding
    assert rec2["records_gathered"] == rec2["detections_delivered"] and not rec2["records_truncated"]
    assert abs(rec2["value"] * rec2["ms_per_step"] - rec1["value"] * rec1["ms_per_step"]) < 1e-4 * rec1["value"] * rec1["ms_per_step"]   # same windows (the line carries 6 digits)


@pytest.mark.gpu
def test_bench_default_list_two_ranks_and_the_memory_it_leaves(tmp_path):
    """The driver's N=2 command with bench.py's DEFAULT `also` list (two ranks on device 0, gloo + the librccl stand-in, config 5 cut to
    32 images): every record arrives with n_gpus 2 and all its detections gathered on rank 0, and each finished workload gives its
    device memory back -- capi's handles had no __del__ once, nine workloads left 160 GB behind and two ranks on one device ran out."""
    import json
    import re
    import subprocess
    import sys
    if not os.path.exists(STUB_RCCL):
        pytest.fail("tests/stub_rccl/librccl_stub.so not built")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FD_DIST_ONE_DEVICE="1", FD_BENCH_DIST_BACKEND="gloo", FD_RCCL_LIB=STUB_RCCL, FD_BENCH_CONFIG5_IMAGES="32", FD_BENCH_MEMLOG="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29577",
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-probe",
           "--full-out", str(tmp_path / "full.json")]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = r.stdout.strip().splitlines()[-1]
    assert len(line) < 8192, len(line)   # the driver could not parse round 5's 22 KB line
    rec = json.loads(line)
    assert set(rec["summary"]) >= {"cascade", "hog_svm", "ffp15", "sdm", "cascade_late", "cascade_group", "config5"} and "also" not in rec
    full = json.load(open(tmp_path / "full.json"))   # the complete records live in the side file
    assert full["value"] == pytest.approx(rec["value"], rel=1e-5)
    recs = [full] + full["also"]
    assert len(recs) == 9 and all(rec["summary"][k]["value"] == pytest.approx(a["value"], rel=1e-5) for k, a in zip(rec["summary"], recs))
    for a in recs:
        assert a["n_gpus"] == 2 and a["value"] > 0, a["config"]["workload"]
        assert a["records_gathered"] == a["detections_delivered"] and not a["records_truncated"], a["config"]["workload"]
    assert recs[-1]["scaling"] == "strong" and all(a["scaling"] == "weak" for a in recs[:-1])
    used = [float(m) for m in re.findall(r"\[mem\] rank 0 after \w+: ([0-9.]+) GB", r.stderr)]
    assert len(used) == 9 and max(used) < 24.0, used   # both ranks' share of one device; 160 GB per rank before the handles freed themselves


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["cascade", "cascade_group", "hog_svm", "ffp15", "sdm"])
def test_bench_line_of_every_workload_with_its_probe(workload, tmp_path):
    """One short run of bench.py per workload WITH its kernel probe and roofline records (the multi-rank test above runs without them):
    the JSON line parses and carries the contract's fields.  (A NameError in one workload's probe once left the driver's default run
    without a line.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--workload", workload, "--also", "none", "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
           "--full-out", str(tmp_path / "full.json")]
    if workload.startswith("cascade"):
        cmd += ["--frames-per-step", "128"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = r.stdout.strip().splitlines()[-1]
    assert len(line) < 8192, len(line)
    rec = json.loads(line)
    full = json.load(open(tmp_path / "full.json"))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in rec, key
    assert rec["value"] > 0 and rec["roofline"]["bound"] in ("hbm", "mfma") and 0 < rec["roofline"]["frac"] <= 1.0
    assert rec["roofline"]["achieved"] > 0 and rec["roofline"]["peak"] > 0 and "workload" in rec["config"]
    if workload == "cascade_group":   # VERDICT r04 task 1c: the heavy-queue profiles name the kernel that dominates THEM
        assert rec["roofline"]["kernel"].startswith("k_wvb") and full["roofline"]["kernel_ms"] > 0.25 * full["roofline"]["cascade_kernels_ms"]
    if workload == "ffp15":
        assert "32 distinct frames" in full["config"]["content"]
    if workload == "sdm":   # 256 distinct crops per batch, >= 4 distinct batches
        assert "4 distinct batches of 256 distinct crops" in full["config"]["content"]
