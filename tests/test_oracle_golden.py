"""Oracle pinned against (a) outputs of the reference's own translation units (tests/golden/ref_*.npz,
generated from /root/reference by tests/golden/make_golden.py) and (b) its own committed regression
vectors.  CPU only."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_vlhog_matches_reference_hog_c(oracle):
    g = np.load(os.path.join(G, "ref_vlhog.npz"))
    for i in range(int(g["n"])):
        cell, nori, var = [int(v) for v in g["par%d" % i]]
        out = oracle.vlhog(g["img%d" % i], cell, nori, var)
        assert out.shape == g["out%d" % i].shape
        assert np.array_equal(out, g["out%d" % i]), "case %d differs from hog.c" % i  # bit-exact


def _sdm_real():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_sdm_real", os.path.join(G, "make_sdm_real.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)   # only defines functions; main() (which reads /root/reference) is not run
    g = np.load(os.path.join(G, "sdm_real_11012014.npz"))
    return g, mod.unpack_model(g)


def test_sdm_real_regressors_oracle(oracle, synth):
    """The reference's one trained model (detect-landmarks/share/models/SDM_Model_HOG_Zhenhua_11012014.txt, re-packed by
    tests/golden/make_sdm_real.py: 22 landmarks, 5 steps, non-adaptive branch of optimize()) through the oracle: the committed per-step
    shapes are reproduced, and the oracle's descriptors at every step equal what the reference's own hog.c produced for them."""
    g, model = _sdm_real()
    L, S = model["L"], model["S"]
    assert [r.shape for r in model["R"]] == [(3169, 44), (3169, 44), (1409, 44), (1409, 44), (353, 44)]
    fb = g["face_box"]
    for f in (0, 17, 31):
        img = synth.make_frame(256, 256, seed=int(g["frame_seed0"]) + f, channels=1)
        st, sh = oracle.sdm_fit(img, model, fb)
        assert st == 0 and np.array_equal(sh, g["oracle_shapes"][f, S])
        if f == 31:
            continue
        for s in range(S):
            nc, cp, nb = [int(v) for v in model["desc_params"][3 * s:3 * s + 3]]
            sh = g["oracle_shapes"][f, s]
            d = oracle.sdm_descriptors(img, sh[:L], sh[L:], 0, variant=1, num_cells=nc, cell_size=cp, num_bins=nb)
            assert d.shape == (L, nc * nc * 16)
            assert np.array_equal(d, g["ref_desc_f%d_s%d" % (f, s)]), (f, s)   # bit-exact against hog.c
    # the landmarks really move at these magnitudes (the synthetic models of config 4 move them by ~1e-2 px)
    mv = np.abs(np.diff(g["oracle_shapes"], axis=1)).max(axis=(0, 2))
    assert mv.min() > 1.0


def test_iimg_matches_reference_iimg_cpp(oracle):
    g = np.load(os.path.join(G, "ref_iimg.npz"))
    for i in range(int(g["n"])):
        for sqr in (0, 1):
            assert np.array_equal(oracle.iimg(g["patch%d" % i], sqr), g["out%d_%d" % (i, sqr)])


def test_svm_kernels_match_reference_libsvm(oracle):
    """libsvm computes everything in fp64; the reference kernels use fp32 SSD/min-sum for float
    inputs, so compare to 1e-5 relative (tolerance stated here)."""
    g = np.load(os.path.join(G, "ref_libsvm.npz"))
    sv, x, coef, rho = g["sv"].astype(np.float32), g["x"].astype(np.float32), g["coef"].astype(np.float32), float(g["rho"])
    for name, kern in (("linear", 0), ("poly", 1), ("rbf", 2), ("hik", 3)):
        kt, deg, gamma, c0 = g["par_" + name]
        m = dict(kernel=kern, dtype=1, sv=sv, coeff=coef, bias=np.float32(rho), threshold=0.0)
        if kern == 1:
            m.update(p0=gamma, p1=c0, p2=deg)   # libsvm poly: (gamma*u'v + coef0)^degree
        elif kern == 2:
            m.update(p0=gamma)
        d = oracle.Svm(m).distance(x)
        ref = g["dec_" + name]
        # rho/coef were rounded to f32 for the reference classes: compare with matching tolerance
        assert np.allclose(d, ref, rtol=2e-5, atol=2e-5 * np.abs(coef).sum()), (name, d, ref)


def test_oracle_live_reference_units(oracle):
    """When oracle/_ref is present (build container, or shipped prebuilt), cross-check on fresh random
    inputs, beyond the committed vectors."""
    if oracle.ref() is None:
        pytest.skip("oracle/_ref/libfdref.so not available")
    rng = np.random.default_rng(123)
    for _ in range(5):
        img = rng.uniform(0, 255, (30, 30)).astype(np.float32)
        assert np.array_equal(oracle.vlhog(img, 10, 9, 1), oracle.ref_vlhog(img, 10, 9, 1))
        img = rng.uniform(0, 255, (27, 33)).astype(np.float32)
        assert np.array_equal(oracle.vlhog(img, 4, 6, 0), oracle.ref_vlhog(img, 4, 6, 0))


def test_cascade_regression_vectors(oracle):
    g = np.load(os.path.join(G, "orc_cascade_160x120.npz"))
    wvm = {k[5:]: g[k] for k in g.files if k.startswith("wvm__")}
    svm = {k[5:]: g[k] for k in g.files if k.startswith("svm__")}
    pyr = oracle.Pyramid(octave_layers=4, min_scale=0.4, max_scale=1.0)
    pyr.update(g["frame"])
    sizes = np.array([[l["index"], l["w"], l["h"]] for l in pyr.layers()], np.int32)
    assert np.array_equal(sizes, g["layer_sizes"])
    sums = np.array([int(pyr.layer(i).astype(np.int64).sum()) for i in range(len(sizes))], np.int64)
    assert np.array_equal(sums, g["layer_sums"])
    assert np.array_equal(pyr.layer(len(sizes) - 1), g["last_layer"])
    w, s = oracle.Wvm(wvm), oracle.Svm(svm)
    pos, lv, fo = oracle.sliding_wvm(pyr, w)
    assert np.array_equal(lv, g["wvm_level"])
    assert np.array_equal(fo, g["wvm_fout"])
    dets, stages = oracle.five_stage(pyr, w, s)
    assert np.array_equal(stages, g["stages"])
    for f in ("cx", "cy", "w", "h", "layer", "lx", "ly"):
        assert np.array_equal(dets[f], g["five"][f])
    pyr2 = oracle.Pyramid(octave_layers=3, min_scale=0.3, max_scale=1.0)
    pyr2.set_layer_filter(1, bins=9)
    pyr2.update(g["frame"])
    _, _, feats = oracle.sliding_hog_svm(pyr2, None, 20, 20, 2, 2, 9, 5, 2, want_feats=10 ** 9)
    assert len(feats) == int(g["hog_n"])
    assert np.array_equal(feats[:64], g["hog_feat_head"])


def test_pyramid_structure_matches_survey_appendix_d(oracle, frame640):
    """ffpDetectApp/FaceFrontal.cfg parameters (float-typed as ffpDetectApp.cpp:407 reads them):
    13 layers 96x72 .. 34x26 and 16,185 windows at step 1."""
    p = oracle.Pyramid(inc=float(np.float32(0.92)), min_scale=float(np.float32(0.05)), max_scale=float(np.float32(0.16)))
    p.update(frame640)
    L = p.layers()
    assert p.octave_layers == 8
    assert [(l["index"], l["w"], l["h"]) for l in L][:3] == [(22, 96, 72), (23, 88, 66), (24, 80, 60)]
    assert (L[-1]["index"], L[-1]["w"], L[-1]["h"]) == (34, 34, 26)
    assert len(p.windows(20, 20, 1, 1)) == 16185
    assert len(p.windows(20, 20, 2, 2)) == 4161


def test_histeq64_properties(oracle):
    rng = np.random.default_rng(0)
    for _ in range(20):
        p = rng.integers(0, 256, (20, 20), dtype=np.uint8)
        e = oracle.histeq64(p)
        assert e.max() in (254, 255)
        order = np.argsort(p.ravel() >> 2, kind="stable")
        assert np.all(np.diff(e.ravel()[order].astype(int)) >= 0)  # monotone in the input bin
    flat = np.full((20, 20), 77, np.uint8)
    assert np.all(oracle.histeq64(flat) == 255)


def test_wvm_invariants(oracle, small_models):
    wvm, _ = small_models
    w = oracle.Wvm(wvm)
    rng = np.random.default_rng(2)
    for _ in range(50):
        patch = rng.integers(0, 256, (20, 20), dtype=np.uint8)
        lv, fo = w.eval(patch)
        assert 0 <= lv < wvm["num_filters"]
        if lv + 1 < wvm["num_filters"]:
            assert fo < wvm["thresholds"][lv]  # early exit means the threshold was missed


def test_empty_and_tiny_inputs(oracle):
    p = oracle.Pyramid(octave_layers=2, min_scale=0.5, max_scale=1.0)
    p.update(np.zeros((10, 10), np.uint8))  # smaller than the patch: no windows
    assert len(p.windows(20, 20, 1, 1)) == 0
    assert len(oracle.overlap_elimination(np.zeros(0, oracle.DET_DTYPE), 5.0, 0.0)) == 0
    assert oracle.block_nms(np.zeros((40, 50), np.float32), 35).sum() == 0


def test_pyramid_histogram_filters_structure(oracle):
    """PyramidHogFilter / SpatialPyramidHistogramFilter restatements against first principles: without
    normalisation the level-l histograms are the sums of their children and the root is the histogram of
    the whole patch; PyramidHog normalises each histogram by 1/sqrt(energy + 1e-4)."""
    rng = np.random.default_rng(3)
    img = np.stack([rng.integers(0, 9, (24, 24)), rng.integers(0, 256, (24, 24))], -1).astype(np.uint8)
    v = oracle.spatial_pyramid_histogram(img, bins=9, levels=3, interpolate=False, normalization=0).reshape(21, 9)
    whole = np.zeros(9, np.float64)
    for b, w in img.reshape(-1, 2):
        whole[b] += np.float32(1.0 / 255.0) * np.float32(w)
    assert np.allclose(v[0], whole, rtol=1e-5)
    lvl1, lvl2 = v[1:5].reshape(2, 2, 9), v[5:21].reshape(4, 4, 9)
    for r in range(2):
        for c in range(2):
            kids = lvl2[2 * r:2 * r + 2, 2 * c:2 * c + 2].reshape(4, 9)
            assert np.array_equal(lvl1[r, c], ((kids[0] + kids[1]) + kids[2]) + kids[3])
    assert np.array_equal(v[0], ((v[1] + v[2]) + v[3]) + v[4])
    # finest level == cell histograms of the spatial filter on the same grid (cell = 24/4 = 6)
    cells = oracle.spatial_histogram(img, bins=9, cell=6, block=1, normalization=0).reshape(16, 9)
    assert np.array_equal(lvl2.reshape(16, 9), cells)
    ph = oracle.pyramid_hog(img, bins=9, levels=3).reshape(21, 9)
    expect = v / np.sqrt((v.astype(np.float32) ** 2).sum(1, dtype=np.float32) + np.float32(1e-4))[:, None]
    assert np.allclose(ph, expect, rtol=2e-6)
    # signed + unsigned: 3/2 * bins values per histogram, energy over the unsigned half
    img8 = img.copy(); img8[..., 0] %= 8
    ps = oracle.pyramid_hog(img8, bins=8, levels=2, signed_and_unsigned=True).reshape(5, 12)
    raw = oracle.spatial_pyramid_histogram(img8, bins=8, levels=2, normalization=0).reshape(5, 8)
    un = raw[:, :4] + raw[:, 4:]
    nrm = 1.0 / np.sqrt((un ** 2).sum(1) + 1e-4)
    assert np.allclose(ps[:, :8], raw * nrm[:, None], rtol=2e-6) and np.allclose(ps[:, 8:], un * nrm[:, None], rtol=2e-6)


def _whi_numpy(x, alpha=1.0, cutoff=0.390625):
    """The "whi" chain through numpy's FFT (independent of the oracle's DFT loops)."""
    h, w = x.shape
    n = w * h
    F = (np.fft.fft2(x.astype(np.float64)) / n).astype(np.complex64)
    rows, cols = (np.arange(h) + h // 2) % h, (np.arange(w) + w // 2) % w
    fx = (np.float32(-0.5) + cols.astype(np.float32) * np.float32(1.0) / np.float32(w - 1)).astype(np.float32)
    fy = (np.float32(-0.5) + rows.astype(np.float32) * np.float32(1.0) / np.float32(h - 1)).astype(np.float32)
    rho = np.sqrt(fx[None, :] ** 2 + fy[:, None] ** 2, dtype=np.float32)
    f = np.power(rho, np.float32(alpha), dtype=np.float32)
    if cutoff > 0:
        f = f * np.exp(-np.power(rho / np.float32(cutoff), 4, dtype=np.float32), dtype=np.float32)
    G = (F.real * f).astype(np.float32) + 1j * (F.imag * f).astype(np.float32)
    # DFT_REAL_OUTPUT consumes the half spectrum only: rebuild the other half by conjugate symmetry
    full = np.empty((h, w), np.complex128)
    for r in range(h):
        for c in range(w):
            rr, cc = (h - r) % h, (w - c) % w
            own = c < cc or (c == cc and r <= rr)
            full[r, c] = G[r, c] if own else np.conj(G[rr, cc])
    val = (np.fft.ifft2(full).real * n).astype(np.float32)
    u8 = np.clip(np.rint(val.astype(np.float64) + 127.0), 0, 255).astype(np.uint8)
    return u8


def test_whi_chain_against_numpy_fft(oracle):
    rng = np.random.default_rng(9)
    flips = total = 0
    for (h, w) in ((20, 20), (24, 24), (16, 32), (15, 21)):
        for _ in range(6):
            x = rng.integers(0, 256, (h, w)).astype(np.uint8)
            x[: h // 2] //= 2
            u8 = _whi_numpy(x)
            eq = oracle.equalize_hist(u8)
            v = eq.astype(np.float32) * np.float32(1.0 / 127.5) + np.float32(-1.0)
            expect = v * np.float32(1.0 / (np.sqrt((v.astype(np.float64) ** 2).sum()) + np.float32(1e-4)))
            got = oracle.whi(x)
            bad = ~np.isclose(got, expect, rtol=1e-5, atol=1e-7)
            flips += int(bad.sum()); total += bad.size
    # a whitened value within float rounding of .5 may round the other way; anything systematic would show up everywhere
    assert flips <= total // 500


def test_rvm_cascade_restatement(oracle, synth):
    """RvmClassifier through the cached evaluation path (RvmClassifier.cpp:94-112): d_0 = -bias + c00 K_0,
    d_k = d_{k-1} + c_kk K_k (the off-diagonal coefficients are never used), against a direct numpy restatement;
    kernels via the SVM oracle (itself pinned against libsvm)."""
    rng = np.random.default_rng(21)
    feats = (rng.integers(0, 256, (400, 16 * 12)).astype(np.float32)) * np.float32(0.5)
    for kernel in (2, 1, 3, 0):
        m = synth.make_rvm(5 + kernel, feats, 16, 12, n_filters=10, kernel=kernel)
        r = oracle.Rvm(m)
        lv, d = r.eval(feats)
        # kernel values from the (libsvm-pinned) SVM oracle: one "SVM" per reduced set vector with coefficient 1, bias 0
        K = np.empty((len(feats), 10))
        for k in range(10):
            s = oracle.Svm(dict(kernel=kernel, dtype=1, sv=m["sv"][k:k + 1], coeff=np.ones(1, np.float32), bias=np.float32(0), p0=m["p0"], p1=m["p1"],
                                p2=m["p2"], threshold=0.0))
            K[:, k] = s.distance(feats)
        for i in range(len(feats)):
            dist, k = -float(m["bias"]), -1
            while True:
                k += 1
                dist = dist + float(m["coeff"][k * (k + 1) // 2 + k]) * K[i, k]
                if not (dist >= float(m["thresholds"][k]) and k + 1 < 10):
                    break
            assert lv[i] == k and d[i] == dist
        assert 0 < (lv == 9).sum() < len(feats)      # the calibrated cascade lets some vectors through and rejects most
        assert r.classify(9, float(m["thresholds"][9]) + 1) and not r.classify(8, 1e9)
        assert r.probability(0.3) == 1.0 / (1.0 + np.exp(m["logistic_a"] + m["logistic_b"] * 0.3))
        r.close()
    # setNumFiltersToUse (RvmClassifier.cpp:119-126)
    m = synth.make_rvm(1, feats, 16, 12, n_filters=6)
    m["num_used"] = 3
    lv, _ = oracle.Rvm(m).eval(feats)
    assert lv.max() == 2


def _np_fhog(img, cs=8, ub=9):
    """independent vectorised FHOG (float64 accumulation) of a gray (h, w) or BGR (h, w, 3) image: hard bin assignment,
    bilinear cell interpolation, alpha 0.2; for BGR the channel with the largest gradient magnitude per pixel"""
    sb = 2 * ub
    rows, cols = img.shape[0] // cs, img.shape[1] // cs
    H, W = rows * cs, cols * cs
    ys, xs = np.arange(H), np.arange(W)

    def grads(I):
        I = I.astype(np.float64)
        dx = ((I[:H][:, np.minimum(xs + 1, img.shape[1] - 1)] - I[:H][:, np.maximum(xs - 1, 0)]).astype(np.float32) / np.float32(510))
        dy = ((I[np.minimum(ys + 1, img.shape[0] - 1)][:, :W] - I[np.maximum(ys - 1, 0)][:, :W]).astype(np.float32) / np.float32(510))
        return dx, dy, np.sqrt(dx * dx + dy * dy, dtype=np.float32)

    if img.ndim == 2:
        dx, dy, magf = grads(img)
    else:
        g = [grads(img[..., c]) for c in range(3)]
        m1, m2, m3 = g[0][2], g[1][2], g[2][2]
        pick = np.where(m1 > m2, np.where(m1 > m3, 0, 2), np.where(m2 > m3, 1, 2))   # FhogFilter.hpp:161-171
        dx = np.choose(pick, [g[0][0], g[1][0], g[2][0]])
        dy = np.choose(pick, [g[0][1], g[1][1], g[2][1]])
        magf = np.choose(pick, [m1, m2, m3])
    mag = magf.astype(np.float64)
    # orientation and hard bin assignment in float32 like the reference: with 18 bins the axis-aligned gradients sit exactly
    # on .5 boundaries, so the bin of a purely vertical gradient is decided by float rounding
    ori = np.arctan2(dy, dx)
    ori[ori < 0] += np.float32(2 * np.pi)
    b = (ori * np.float32(sb / np.float32(2 * np.pi)) + np.float32(0.5)).astype(int) % sb
    hist = np.zeros((rows, cols, sb))

    def coeff(n, cnt):
        real = (np.arange(n) + 0.5) / cs - 0.5
        i1 = np.floor(real).astype(int); i2 = i1 + 1
        w2 = real - i1; w1 = i2 - real
        lo, hi = i1 < 0, i2 >= cnt
        i1[lo] = i2[lo]; w1[lo] = 0
        i2[hi] = i1[hi]; w2[hi] = 0
        return i1, i2, w1, w2
    r1, r2, rw1, rw2 = coeff(H, rows)
    c1, c2, cw1, cw2 = coeff(W, cols)
    for (ri, rw) in ((r1, rw1), (r2, rw2)):
        for (ci, cw) in ((c1, cw1), (c2, cw2)):
            np.add.at(hist, (ri[:, None].repeat(W, 1), ci[None, :].repeat(H, 0), b), mag * rw[:, None] * cw[None, :])
    u = hist[..., :ub] + hist[..., ub:]
    E = (u ** 2).sum(-1)
    Ep = np.pad(E, 1, mode="edge")

    def blk(dr, dc):
        return Ep[dr:dr + rows, dc:dc + cols] + Ep[dr:dr + rows, dc + 1:dc + 1 + cols] + Ep[dr + 1:dr + 1 + rows, dc:dc + cols] + Ep[dr + 1:dr + 1 + rows, dc + 1:dc + 1 + cols]
    n = [1.0 / np.sqrt(blk(dr, dc) + 1e-4) for dr in (0, 1) for dc in (0, 1)]
    out = np.zeros((rows, cols, 3 * ub + 4))
    for k in range(4):
        vs = np.minimum(0.2, hist * n[k][..., None])
        out[..., :sb] += 0.5 * vs
        out[..., sb:sb + ub] += 0.5 * np.minimum(0.2, u * n[k][..., None])
        out[..., sb + ub + k] = 0.2357 * vs.sum(-1)
    return out


def test_fhog_against_numpy_restatement(oracle):
    """FhogFilter + FhogAggregationFilter restatement against an independent vectorised numpy version (float64 accumulation,
    so 1e-5): default parameters (cell 8, 9 unsigned bins, hard bin assignment, bilinear cell interpolation, alpha 0.2), on a
    CV_8UC1 and on a CV_8UC3 image (per pixel the channel with the largest gradient magnitude)."""
    rng = np.random.default_rng(6)
    img = rng.integers(0, 256, (67, 90)).astype(np.uint8)
    img[20:40, 30:70] //= 3
    out = _np_fhog(img)
    got = oracle.fhog(img)
    assert got.shape == out.shape
    assert np.allclose(got, out, rtol=2e-4, atol=2e-5)
    bgr = rng.integers(0, 256, (67, 90, 3)).astype(np.uint8)
    bgr[10:30, 20:60, 1] //= 4
    bgr[..., 2] = (bgr[..., 2].astype(int) * 3 // 4).astype(np.uint8)
    gotc = oracle.fhog(bgr)
    assert np.allclose(gotc, _np_fhog(bgr), rtol=2e-4, atol=2e-5)
    assert not np.allclose(gotc, oracle.fhog(bgr[..., 0].copy()), atol=1e-3)
    assert np.array_equal(oracle.fhog(np.repeat(img[:, :, None], 3, 2)), got)   # equal channels: ties fall through to channel 3
    for args in ((8, 9, False, False), (4, 6, False, True)):   # other cell sizes / bin counts, shape + range sanity
        f = oracle.fhog(img, *args)
        assert f.shape == (img.shape[0] // args[0], img.shape[1] // args[0], 3 * args[1] + 4)
        assert f.min() >= 0 and f[..., :3 * args[1]].max() <= 0.4 + 1e-6


def _rvm_from_fixture(g):
    m = {k[5:]: g[k] for k in g.files if k.startswith("rvm__")}
    for k in ("kernel", "filter_w", "filter_h", "num_used"):
        m[k] = int(m[k])
    for k in ("p0", "p1", "p2", "logistic_a", "logistic_b", "bias"):
        m[k] = float(m[k])
    return m


def test_next_rows_regression_vectors(oracle):
    """tests/golden/orc_next_rows_128x96.npz (FHOG on gray / BGR images, aggregated detector, cascaded RVM, whitening chain,
    interpolating HogFilter): the oracle must keep reproducing its committed outputs."""
    g = np.load(os.path.join(G, "orc_next_rows_128x96.npz"))
    frame = g["frame"]
    gray = oracle.bgr2gray(frame)
    assert np.array_equal(gray, g["gray"])
    assert np.array_equal(oracle.fhog(gray, 8, 9, False, True, 0.2), g["fhog_gray"])
    assert np.array_equal(oracle.fhog(frame, 8, 9, False, True, 0.2), g["fhog_bgr"])
    assert np.array_equal(oracle.fhog(gray, 4, 6, True, False, 0.2), g["fhog_gray_c4_b6_ib"])
    sc, cc = oracle.aggregated_candidates(frame, g["agg_weights"], 0.05, float(g["agg_threshold"]), cell_size=8, octave_layers=4)
    assert np.array_equal(sc, g["agg_cand_scores"]) and np.array_equal(np.asarray(cc).reshape(-1, 4), g["agg_cand_boxes"])
    fs, fb = oracle.nms_iou(sc, cc, 0.3, 0)
    assert np.array_equal(fs, g["agg_final_scores"]) and np.array_equal(np.asarray(fb).reshape(-1, 4), g["agg_final_boxes"])
    pyr = oracle.Pyramid(octave_layers=2, min_scale=0.4, max_scale=0.8)
    pyr.update(frame)
    layers = [pyr.layer(i) for i in range(len(pyr.layers()))]
    pat = np.stack([oracle.histeq64(np.ascontiguousarray(layers[lp][ly:ly + 20, lx:lx + 20])) for lp, lx, ly, *_ in pyr.windows(20, 20, 2, 2)])
    lo, do = oracle.Rvm(_rvm_from_fixture(g)).eval(pat.reshape(len(pat), -1).astype(np.float32))
    assert np.array_equal(lo, g["rvm_level"]) and np.array_equal(do, g["rvm_dist"])
    assert np.array_equal(np.stack([oracle.whi(q, 1.0, 0.390625) for q in g["whi_patches"]]), g["whi_out"])
    assert np.array_equal(np.stack([oracle.equalize_hist(q) for q in g["whi_patches"]]), g["eqhist_out"])


def _np_reflect101(p, n):
    """cv::borderInterpolate(p, n, BORDER_REFLECT_101)"""
    if n == 1:
        return 0
    while p < 0 or p >= n:
        p = -p if p < 0 else 2 * n - 2 - p
    return p


def _np_pyrdown(img):
    """cv::pyrDown, 8U: separable [1 4 6 4 1], BORDER_REFLECT_101, (sum + 128) >> 8 -- written from the definition, rows first"""
    h, w = img.shape
    dh, dw = (h + 1) // 2, (w + 1) // 2
    k = (1, 4, 6, 4, 1)
    a = img.astype(np.int64)
    hor = np.zeros((h, dw), np.int64)
    for dx in range(dw):
        for t in range(5):
            hor[:, dx] += k[t] * a[:, _np_reflect101(2 * dx + t - 2, w)]
    out = np.zeros((dh, dw), np.int64)
    for dy in range(dh):
        for t in range(5):
            out[dy] += k[t] * hor[_np_reflect101(2 * dy + t - 2, h)]
    return ((out + 128) >> 8).astype(np.uint8)


def _np_resize_linear_u8(img, dw, dh):
    """cv::resize(INTER_LINEAR) for 8UC1 as OpenCV 2.4 computes it: float coordinates from double scales, 11-bit coefficients rounded
    half-to-even, horizontal pass in int32, vertical pass (b * (r >> 4)) >> 16, + 2, >> 2 -- vectorised, independent of oracle/orc_image.cpp"""
    sh, sw = img.shape

    def axis(dn, sn, clampCoeff):
        scale = 1.0 / (dn / float(sn))
        f = ((np.arange(dn) + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        if clampCoeff:
            lo, hi = s < 0, s >= sn - 1
            f[lo | hi] = 0
            s[lo] = 0
            s[hi] = sn - 1
        c0 = np.rint((np.float32(1) - f).astype(np.float32) * np.float32(2048)).astype(np.int64)
        c1 = np.rint(f * np.float32(2048)).astype(np.int64)
        return s, c0, c1
    xs, a0, a1 = axis(dw, sw, True)
    ys, b0, b1 = axis(dh, sh, False)
    a = img.astype(np.int64)
    x1 = np.minimum(xs + 1, sw - 1)
    hor = a[:, xs] * a0[None, :] + a[:, x1] * a1[None, :]   # [sh, dw]
    r0 = hor[np.clip(ys, 0, sh - 1)]
    r1 = hor[np.clip(ys + 1, 0, sh - 1)]
    v = (((b0[:, None] * (r0 >> 4)) >> 16) + ((b1[:, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return v.astype(np.uint8)


def test_opencv_primitives_against_numpy_restatement(oracle):
    """cvtColor(BGR2GRAY), cv::resize(INTER_LINEAR) and cv::pyrDown on 8-bit images: the C++ restatement (oracle/orc_image.cpp, what the
    HIP pyramid kernels are compared with) against a second, independently written numpy restatement of the OpenCV 2.4 arithmetic --
    including images of one to three pixels across (multiple border reflections) and up- as well as down-scaling."""
    rng = np.random.default_rng(77)
    bgr = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    want = ((bgr[..., 0].astype(np.int64) * 1868 + bgr[..., 1].astype(np.int64) * 9617 + bgr[..., 2].astype(np.int64) * 4899 + 8192) >> 14).astype(np.uint8)
    assert np.array_equal(oracle.bgr2gray(bgr), want)
    for (w, h) in ((53, 37), (64, 48), (5, 4), (3, 3), (2, 7), (1, 5), (9, 1), (1, 1), (131, 2)):
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        assert np.array_equal(oracle.pyrdown(img), _np_pyrdown(img)), ("pyrDown", w, h)
    for (w, h, dw, dh) in ((640, 480, 589, 442), (97, 81, 50, 41), (97, 81, 96, 80), (33, 21, 33, 21), (20, 20, 31, 29), (7, 5, 3, 2),
                           (300, 200, 151, 199), (5, 3, 9, 7), (2, 2, 1, 1)):
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        assert np.array_equal(oracle.resize_linear_u8(img, dw, dh), _np_resize_linear_u8(img, dw, dh)), ("resize", w, h, dw, dh)


def test_histeq64_against_numpy_restatement(oracle, synth):
    """HistEq64Filter.cpp:32-125 (64 bins of the pixel >> 2, fp32 pdf and sequential fp32 cdf, (uchar)floor(cdf + 0.5)) restated twice:
    the C++ oracle against a vectorised numpy form, on patches of every cfg-implied size, flat and two-valued ones included (exact .5 ties)."""
    rng = np.random.default_rng(123)
    for (pw, ph) in ((20, 20), (24, 24), (16, 24), (32, 16), (32, 24), (19, 21), (7, 5)):
        pats = rng.integers(0, 256, (64, ph, pw), dtype=np.uint8)
        pats[0] = 200
        pats[1, : ph // 2] = 3
        pats[1, ph // 2:] = 251
        pats[2] = (np.arange(ph * pw).reshape(ph, pw) * 40 // (ph * pw) * 4).astype(np.uint8)   # 40 bins, equal counts when 40 | d: ties at .5
        want = synth.histeq64_np(pats)
        for i in range(len(pats)):
            assert np.array_equal(oracle.histeq64(pats[i]), want[i]), (pw, ph, i)


def _np_wvm_eval(m, patch):
    """WvmClassifier.cpp:100-149,191-346 + IImg.cpp:26-65 written a second time, in numpy scalars with the reference's types (float
    integral images, double sum_xp / norm, float kernel values and level sums); independent of oracle/orc_classify.cpp."""
    f32, f64 = np.float32, np.float64
    fw, fh, F, NP = int(m["filter_w"]), int(m["filter_h"]), int(m["num_filters"]), int(m["num_per_level"])
    nu = int(m["num_used"])
    nu = F if (nu > F or nu == 0) else nu
    p = patch.astype(np.int64)
    ii = np.zeros((fh + 1, fw + 1), np.int64)
    ii[1:, 1:] = p.cumsum(0).cumsum(1)              # every partial sum is an integer below 2^24: exact in the reference's floats
    total = f32(0)                                   # the squared image's bottom-right entry: float row sums added row by row
    for r in range(fh):
        rowsum = f32(0)
        for c in range(fw):
            rowsum = f32(rowsum + f32(int(p[r, c]) * int(p[r, c])))
        total = f32(total + rowsum)
    basis, bias = f32(m["basis_param"]), f32(m["bias"])
    u = np.zeros(NP, np.float32)
    out = np.zeros(F, np.float32)
    level = -1
    while True:
        level += 1
        n = level % NP
        v0, cnt = int(m["val_off"][level]), int(m["val_off"][level + 1] - m["val_off"][level])
        sumv0 = f32(ii[fh, fw])
        sum_xp = f64(0)
        for v in range(1, cnt):
            sv = 0
            for r in range(int(m["rec_off"][v0 + v]), int(m["rec_off"][v0 + v + 1])):
                x1, y1, x2, y2 = (int(q) for q in m["rects"][r])
                sv += int(ii[y2 + 1, x2 + 1] - ii[y1, x2 + 1] - ii[y2 + 1, x1] + ii[y1, x1])
            sumv0 = f32(sumv0 - f32(sv))
            sum_xp = f64(sum_xp + f64(f32(sv)) * f64(m["val"][v0 + v]))
        sum_xp = f64(sum_xp + f64(sumv0) * f64(m["val"][v0]))
        sum_xp = f64(sum_xp + f64(u[n]))
        u[n] = f32(sum_xp)
        norm = f64(total)
        norm = f64(norm - f64(2) * sum_xp)
        norm = f64(norm + f64(m["pp"][level]))
        out[level] = f32(np.exp(f64(-basis) * norm))
        res = f32(-bias)
        w = m["hk_weights"].reshape(F, F)[level]
        for q in range(level + 1):
            res = f32(res + f32(f32(w[q]) * out[q]))
        if not (res >= f32(m["thresholds"][level]) and level + 1 < nu):
            return level, res


def test_wvm_cascade_against_numpy_restatement(oracle, synth, small_models, frame640):
    """The C++ restatement of the WVM cascade (what every window's level and fp32 output on the GPU is compared with) against a second
    restatement in numpy scalars: same last level and bit-identical fp32 output on equalised patches, for the full model and for
    numUsedFilters between level groups."""
    wvm = small_models[0]
    gray = oracle.bgr2gray(frame640)
    rng = np.random.default_rng(31)
    pats = synth.histeq64_np(synth.random_patches(gray[::4, ::4].copy(), 20, 20, 120, rng))
    pats[0] = 255
    pats[1] = 0
    for nu in (wvm["num_used"], 17):
        m = dict(wvm)
        m["num_used"] = nu
        wo = oracle.Wvm(m)
        deep = 0
        for p in pats:
            lo, fo = wo.eval(p)
            ln, fn = _np_wvm_eval(m, p)
            assert lo == ln and np.float32(fo).tobytes() == np.float32(fn).tobytes(), (nu, lo, ln, fo, fn)
            deep += lo >= 6
        assert deep >= 5   # the comparison is not only about first-level rejects


def _np_hist_cache(size, count):
    """HistogramFilter::createCache (HistogramFilter.cpp:198-219): the two cells a pixel row / column feeds and their float weights"""
    f32 = np.float32
    out = []
    for i in range(size):
        real = float(count) * (float(i) + 0.5) / float(size) - 0.5
        i1 = int(np.floor(real))
        i2 = i1 + 1
        w2 = f32(real - i1)
        w1 = f32(f32(1) - w2)
        if i1 < 0:
            i1, w1 = i2, f32(0)
        elif i2 >= count:
            i2, w2 = i1, f32(0)
        out.append((i1, i2, w1, w2))
    return out


def _np_hog_filter(binimg, bins, cw, chh, bw, bh, sau, interpolate=False):
    """HogFilter::applyTo without cell interpolation on a (bin, weight) image: HistogramFilter.cpp:140-175 (cell bounds by integer
    division, histogram[bin] += factor * weight in scan order), HogFilter.cpp:102-122 (cell energies), :66-100 (block normalisation and
    output order) -- a second restatement in numpy float32 scalars, written from the reference's source."""
    f32 = np.float32
    h, w = binimg.shape[:2]
    rows, cols = int(np.rint(h / chh)), int(np.rint(w / cw))   # cvRound (these quotients are never at .5 in the test)
    factor = f32(1.0) / f32(255.0)
    hist = np.zeros((rows, cols, bins), np.float32)
    if interpolate:   # HistogramFilter.cpp:66-99: every pixel feeds up to four cells, (weight * row weight) * column weight
        rc, ccache = _np_hist_cache(h, rows), _np_hist_cache(w, cols)
        for y in range(h):
            r0, r1, rw0, rw1 = rc[y]
            for x in range(w):
                b, wt = int(binimg[y, x, 0]), f32(factor * f32(int(binimg[y, x, 1])))
                c0, c1, cw0, cw1 = ccache[x]
                for (ri, rwt, ok_r) in ((r0, rw0, r0 >= 0), (r1, rw1, r1 < rows)):
                    for (ci, cwt, ok_c) in ((c0, cw0, c0 >= 0), (c1, cw1, c1 < cols)):
                        if ok_r and ok_c:
                            hist[ri, ci, b] = f32(hist[ri, ci, b] + f32(f32(wt * rwt) * cwt))
    else:
      for cr in range(rows):
        for cc in range(cols):
            for y in range((cr * h) // rows, ((cr + 1) * h) // rows):
                for x in range((cc * w) // cols, ((cc + 1) * w) // cols):
                    b, wt = int(binimg[y, x, 0]), int(binimg[y, x, 1])
                    hist[cr, cc, b] = f32(hist[cr, cc, b] + f32(factor * f32(wt)))
    half = bins // 2
    energy = np.zeros((rows, cols), np.float32)
    for cr in range(rows):
        for cc in range(cols):
            e = f32(0)
            if sau:
                for b in range(half):
                    u = f32(hist[cr, cc, b] + hist[cr, cc, half + b])
                    e = f32(e + f32(u * u))
            else:
                for b in range(bins):
                    e = f32(e + f32(hist[cr, cc, b] * hist[cr, cc, b]))
            energy[cr, cc] = e
    out = []
    for br in range(rows - bh + 1):
        for bc in range(cols - bw + 1):
            e = f32(0)
            for cr in range(br, br + bh):
                for cc in range(bc, bc + bw):
                    e = f32(e + energy[cr, cc])
            nrm = f32(f32(1) / f32(np.sqrt(f32(e + f32(1e-4)))))
            for cr in range(br, br + bh):
                for cc in range(bc, bc + bw):
                    out += [f32(nrm * hist[cr, cc, b]) for b in range(bins)]
                    if sau:
                        out += [f32(nrm * f32(hist[cr, cc, b] + hist[cr, cc, half + b])) for b in range(half)]
    return np.asarray(out, np.float32)


def test_hog_filter_against_numpy_restatement(oracle):
    """HistogramFilter / HogFilter (rows a23, a24) restated twice: oracle/orc_features.cpp against numpy float32 scalars, bit for bit,
    on (bin, weight) patches of the config-2 shape and of shapes whose cells do not divide the patch, without and with the bilinear
    interpolation between cells (the pixel order of the interpolating form matters: four cells per pixel, row-major pixels)."""
    rng = np.random.default_rng(9)
    for (w, h, bins, cw, chh, bw, bh, sau) in ((20, 20, 9, 5, 5, 2, 2, False), (20, 20, 8, 5, 5, 2, 2, True), (24, 16, 6, 5, 4, 2, 1, False),
                                                (19, 21, 9, 6, 5, 1, 2, False), (32, 24, 12, 8, 8, 3, 2, True)):
        img = np.zeros((h, w, 2), np.uint8)
        img[..., 0] = rng.integers(0, bins, (h, w))
        img[..., 1] = rng.integers(0, 256, (h, w))
        for interp in (False, True):
            got = oracle.hog_filter(img, bins, cw, bw, interpolate=interp, signed_and_unsigned=sau, cell_h=chh, block_h=bh)
            want = _np_hog_filter(img, bins, cw, chh, bw, bh, sau, interp)
            assert got.shape == want.shape and got.tobytes() == want.tobytes(), (w, h, bins, cw, chh, bw, bh, sau, interp, np.abs(got - want).max())


def test_overlap_elimination_against_python_restatement(oracle):
    """OverlapElimination.cpp:44-105 restated twice: the C++ oracle against the reference's loop written in Python (sort by probability,
    descending; an element survives unless an earlier survivor lies within d in x and y and the width ratio exceeds `ratio`; d = dist
    pixels above 1, dist x the larger width otherwise; ratio outside (0, 1] counts as 0).  Distinct probabilities: the reference's
    order of equal ones is whatever its std::sort leaves."""
    from oracle.pyoracle import DET_DTYPE
    rng = np.random.default_rng(4)
    for n in (1, 2, 40, 500):
        d = np.zeros(n, DET_DTYPE)
        d["cx"] = rng.integers(-20, 300, n)
        d["cy"] = rng.integers(-20, 200, n)
        d["w"] = rng.choice([20, 22, 25, 31, 40, 63], n)
        d["h"] = d["w"]
        d["prob"] = rng.permutation(n) / float(n) * 0.9 + 0.05
        for dist, ratio in ((5.0, 0.0), (0.5, 0.0), (1.0, 0.0), (20.0, 0.8), (12.5, 1.0), (3.0, 1.5), (3.0, -1.0)):
            r = np.float32(ratio) if 0.0 < ratio <= 1.0 else np.float32(0)
            order = sorted(range(n), key=lambda i: -d["prob"][i])
            keep = []
            for i in order:
                ok = True
                for a in keep:
                    dd = np.float32(dist) * np.float32(max(d["w"][a], d["w"][i])) if dist <= 1.0 else np.float32(dist)
                    if (abs(int(d["cx"][a]) - int(d["cx"][i])) < dd and abs(int(d["cy"][a]) - int(d["cy"][i])) < dd
                            and np.float32(min(d["w"][a], d["w"][i])) / np.float32(max(d["w"][a], d["w"][i])) > r):
                        ok = False
                        break
                if ok:
                    keep.append(i)
            assert list(oracle.overlap_elimination(d, dist, ratio)) == keep, (n, dist, ratio)


def test_block_nms_against_numpy_restatement(oracle):
    """nonMaximaSuppression (FiveStageSlidingWindowDetector.cpp:143-184) restated twice, unmasked and masked maps whose selections are
    never empty (what cv::minMaxLoc returns for an empty selection is an OpenCV property both restatements would only assume): a block's
    first maximum (row-major) is a local maximum iff it exceeds the maximum of its (2 sz + 1)^2 neighbourhood outside the block."""
    rng = np.random.default_rng(12)
    for (M, N, sz) in ((40, 50, 3), (37, 29, 5), (64, 64, 7), (20, 33, 1), (90, 70, 35)):
        src = rng.random((M, N)).astype(np.float32) + np.float32(0.01)
        src[rng.random((M, N)) < 0.1] = np.float32(0.5)   # equal values: the first one in scan order is the block's candidate
        for masked in (False, True):
            mask = None
            if masked:
                mask = np.where(rng.random((M, N)) < 0.8, 255, 0).astype(np.uint8)
            want = np.zeros((M, N), np.uint8)
            skip = False
            for m in range(0, M, sz + 1):
                for n in range(0, N, sz + 1):
                    i1, j1 = min(m + sz + 1, M), min(n + sz + 1, N)
                    blk = src[m:i1, n:j1].astype(np.float64)
                    sel = np.ones(blk.shape, bool) if mask is None else mask[m:i1, n:j1] != 0
                    if not sel.any():
                        skip = True
                        continue
                    v = np.where(sel, blk, -np.inf)
                    k = int(np.argmax(v))   # first occurrence, row-major
                    cy, cx = m + k // blk.shape[1], n + k % blk.shape[1]
                    a0, a1, b0, b1 = max(cy - sz, 0), min(cy + sz + 1, M), max(cx - sz, 0), min(cx + sz + 1, N)
                    nb = src[a0:a1, b0:b1].astype(np.float64)
                    ns = np.ones(nb.shape, bool) if mask is None else mask[a0:a1, b0:b1] != 0
                    r0, c0 = m - a0, n - b0
                    ns[r0:min(r0 + sz + 1, nb.shape[0]), c0:min(c0 + sz + 1, nb.shape[1])] = False
                    if not ns.any():
                        skip = True
                        continue
                    if v.max() > np.where(ns, nb, -np.inf).max():
                        want[cy, cx] = 255
            if skip:
                continue
            assert np.array_equal(oracle.block_nms(src, sz, mask), want), (M, N, sz, masked)


def test_greyworld_against_numpy_restatement(oracle):
    """GreyWorldNormalizationFilter.cpp:20-71 (continuous image) restated twice: channel sums and maxima, the largest max / mean sets the
    common scale, cvRound (half to even) and saturation -- numpy doubles against the C++ oracle."""
    rng = np.random.default_rng(21)
    for shape, hi in (((37, 53, 3), 256), ((8, 8, 3), 40), ((20, 31, 3), 200)):
        img = rng.integers(0, hi, shape, dtype=np.uint8)
        img[0, 0] = (hi - 1, 1, 3)
        n = shape[0] * shape[1]
        mean = img.reshape(-1, 3).astype(np.float64).sum(0) / n
        mx = (img.reshape(-1, 3).max(0).astype(np.float64) / mean).max()
        scale = 255.0 / (mean * mx)
        want = np.clip(np.rint(scale[None, None, :] * img.astype(np.float64)), 0, 255).astype(np.uint8)
        assert np.array_equal(oracle.greyworld(img), want), shape


def test_gradient_filter_and_binning_against_numpy_restatement(oracle):
    """GradientFilter.cpp:38-59 (cv::Sobel with ksize 1 / 3, scale 1/2 / 1/8, delta 127, BORDER_REFLECT_101, 8-bit saturation) and
    GradientBinningFilter.cpp:18-93 (the 65536-entry look-up table from atan2 / sqrt in double) restated twice, numpy against the C++ oracle."""
    import math
    rng = np.random.default_rng(17)
    img = rng.integers(0, 256, (23, 31), dtype=np.uint8)
    a = img.astype(np.int64)

    def refl(p, n):
        return _np_reflect101(p, n)
    H, W = a.shape
    for ksize in (1, 3):
        gx = np.zeros_like(a)
        gy = np.zeros_like(a)
        for y in range(H):
            for x in range(W):
                xm, xp, ym, yp = refl(x - 1, W), refl(x + 1, W), refl(y - 1, H), refl(y + 1, H)
                if ksize == 1:
                    gx[y, x] = a[y, xp] - a[y, xm]
                    gy[y, x] = a[yp, x] - a[ym, x]
                else:
                    gx[y, x] = (a[ym, xp] - a[ym, xm]) + 2 * (a[y, xp] - a[y, xm]) + (a[yp, xp] - a[yp, xm])
                    gy[y, x] = (a[yp, xm] - a[ym, xm]) + 2 * (a[yp, x] - a[ym, x]) + (a[yp, xp] - a[ym, xp])
        scale = 0.5 if ksize == 1 else 1.0 / 8
        want = np.stack([np.clip(np.rint(gx * scale + 127), 0, 255), np.clip(np.rint(gy * scale + 127), 0, 255)], -1).astype(np.uint8)
        got = oracle.gradient_filter(img, ksize, 0)
        assert np.array_equal(got, want), ksize
    grad = rng.integers(0, 256, (40, 40, 2), dtype=np.uint8)
    grad[0, :16, 0] = 127   # zero x gradient: atan2(+-y, 0)
    grad[1, :16, 1] = 127
    grad[2, 0] = (127, 127)
    for bins, signed in ((9, False), (8, True), (12, True), (6, False)):
        for interp in (False, True):
            want = np.zeros((40, 40, 4 if interp else 2), np.uint8)
            for y in range(40):
                for x in range(40):
                    gxv, gyv = (float(grad[y, x, 0]) - 127) / 255, (float(grad[y, x, 1]) - 127) / 255
                    d = math.atan2(gyv, gxv)
                    mag = math.sqrt(gxv * gxv + gyv * gyv)
                    if signed:
                        b = (d + math.pi) * bins / (2 * math.pi)
                    else:
                        b = (d + math.pi if d < 0 else d) * bins / math.pi
                    sat = lambda v: int(min(255, max(0, np.rint(v))))
                    fl = math.floor(b)
                    if interp:
                        w3 = sat(255 * mag * (b - fl))
                        want[y, x] = (int(fl) % 256 % bins, sat(255 * mag - w3), int(math.ceil(b)) % 256 % bins, w3)
                    else:
                        want[y, x] = (int(fl + (1 if b - fl >= 0.5 else 0)) % 256 % bins, sat(255 * mag))
            got = oracle.gradient_binning(grad, bins, signed, interp)
            assert np.array_equal(got, want), (bins, signed, interp)


def test_lbp_against_numpy_restatement(oracle):
    """LbpFilter.cpp:20-85 + LbpFilter.hpp:88-178 restated twice: the three neighbourhood codes with BORDER_REPLICATE, and the uniform
    map (patterns with at most two 0/1 transitions get 1..58 in code order, the rest 0)."""
    rng = np.random.default_rng(23)
    img = rng.integers(0, 256, (19, 27), dtype=np.uint8)
    img[5:9, 5:9] = 77   # ties: strictly greater only
    p = np.pad(img.astype(np.int64), 1, mode="edge")
    H, W = img.shape
    c = p[1:-1, 1:-1]
    nb = lambda dy, dx: (p[1 + dy:1 + dy + H, 1 + dx:1 + dx + W] > c).astype(np.int64)
    lbp8 = (nb(-1, -1) << 7) | (nb(-1, 0) << 6) | (nb(-1, 1) << 5) | (nb(0, 1) << 4) | (nb(1, 1) << 3) | (nb(1, 0) << 2) | (nb(1, -1) << 1) | nb(0, -1)
    lbp4 = (nb(-1, 0) << 3) | (nb(0, 1) << 2) | (nb(1, 0) << 1) | nb(0, -1)
    lbp4r = (nb(-1, -1) << 3) | (nb(-1, 1) << 2) | (nb(1, 1) << 1) | nb(1, -1)
    umap, nxt = np.zeros(256, np.int64), 1
    for code in range(256):
        bits = [(code >> k) & 1 for k in range(8)]
        prev, tr = bits[7], 0
        for b in bits:
            if b != prev:
                tr, prev = tr + 1, b
        if tr <= 2:
            umap[code], nxt = nxt, nxt + 1
    assert nxt == 59
    for lbp_type, want in ((0, lbp8), (1, umap[lbp8]), (2, lbp4), (3, lbp4r)):
        assert np.array_equal(oracle.lbp(img, lbp_type), want.astype(np.uint8)), lbp_type
