"""The reference-shaped C++ host layer (featuredetection_amd/host) driven through its example apps:
ffp_detect_app mirrors ffpDetectApp's object graph (config 1 plumbing), sdm_fit_app the two SDM calls.
Their printed results must equal the oracle's."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "featuredetection_amd")


def _run(args):
    env = dict(os.environ, LD_LIBRARY_PATH=PKG + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run(args, capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    return r.stdout


def test_ffp_detect_app_matches_oracle(tmp_path, oracle, synth, frame640, small_models):
    app = os.path.join(PKG, "ffp_detect_app")
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
""" % (tmp_path / "face.fdwvm", tmp_path / "face.svm.txt")
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
